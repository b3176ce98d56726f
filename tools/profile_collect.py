#!/usr/bin/env python3
"""tools/profile_collect.py <tag> -- copy what is to be judged from gpurun_out/prof_<tag>/ (written by
tools/profile_default.sh <tag> pmc) into profiles/ and derive the two small files bench.py reads (they carry the
hash of the library sources they were measured on; bench.py attaches them only to a run of the same build):
  profiles/r03_bound.json    counter-based bound blocks: the C2 headline kernel, the bucket-major kernel of the 1B leg
  profiles/r03_traffic.json  memory-side bytes per launch of the 1B streaming kernel (roofline.traffic)
Counter conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section): SQ_*_CYCLES and SQ_WAIT_* /
SQ_ACTIVE_* are quad-cycles summed over the SQs; FETCH_SIZE / WRITE_SIZE are KiB, and FETCH_SIZE is
doubled for kernels whose loads are 16 B per lane (the gfx950 rule)."""
import json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vaq_amd import build
HASH = build.source_hash()
R = "r03"
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
C2 = "scan_bytes_bf_kernel<8, true>"
STREAM = "scan_bytes_inplace_kernel<16, 2, true>"

shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, R + "_default_kernel_stats.csv"))
shutil.copy(os.path.join(src, "kt_bench.json"), os.path.join(dst, R + "_default_bench_under_rocprof.json"))
vals = {}
keep = []
for line in open(os.path.join(src, "pmc_summary.txt")):
    f = line.rstrip("\n").split("\t")
    if len(f) < 4 or "scan_" not in f[0]:
        continue
    keep.append(line)
    name = f[0].replace("void vaq::", "").split("(")[0]
    vals.setdefault(name, {})[f[1]] = float(f[2].split("=")[1])
open(os.path.join(dst, R + "_default_pmc_scan.txt"), "w").writelines(keep)

c = vals[C2]
n_simd = 256 * 4
cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
waves = c["SQ_WAVES"]
bound = {"c2": {
    "kernel": C2,
    "bound": "instruction issue + latency of the per-workgroup phases (cache-resident, bucket-pruned scan: no HBM "
             "roofline applies; VALU busy %.2f, %.1f of 8 resident waves per SIMD on average, waves parked %.2f of "
             "their life)" % (c["SQ_ACTIVE_INST_VALU"] * 4 / (n_simd * cycles), c["SQ_WAVE_CYCLES"] * 4 / (n_simd * cycles),
                              c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]),
    "source": "profiles/r03_default_pmc_scan.txt (rocprofv3 --pmc passes of `python3 bench.py --steps 20 "
              "--warmup 5`, tools/profile_default.sh + tools/profile_collect.py)",
    "kernel_cycles": round(cycles),
    "waves": int(waves),
    "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / waves),
    "salu_insts_per_wave": round(c["SQ_INSTS_SALU"] / waves),
    "lds_insts_per_wave": round(c["SQ_INSTS_LDS"] / waves),
    "vmem_read_insts_per_wave": round(c["SQ_INSTS_VMEM_RD"] / waves),
    "valu_busy_frac": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (n_simd * cycles), 3),
    "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3),
    "wave_wait_share": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
    "issue_stall_share": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
    "mean_waves_per_simd": round(c["SQ_WAVE_CYCLES"] * 4 / (n_simd * cycles), 2),
    "memory_side_bytes_per_launch": (c["FETCH_SIZE"] * 2 + c["WRITE_SIZE"]) * 1024,
    "note": "SQ_* cycle counters are quad-cycles; FETCH_SIZE is doubled per the gfx950 rule for 16-B/lane loads "
            "(MI355X_MICROARCH.md, HBM).  Earlier kernels for the same workload: round 1 (scan_bytes_kernel<8,1,1>, "
            "1.47 ms) 12.6 k VALU + 10.7 k SALU per wave, VALU busy 0.56, wait share 0.56, 5.3 waves per SIMD; round 2 "
            "before the prefetch fix (1.00 ms) 6.5 k + 6.4 k, VALU busy 0.42, wait share 0.56, 3.5 waves per SIMD; with "
            "the prefetch fix but before expensive queries were dispatched first (0.72 ms) VALU busy 0.42, 4.2 waves "
            "per SIMD; end of round 2 / start of round 3 (0.534 ms, 7 workgroups per CU) 4.9 k VALU + 5.5 k SALU + 1.0 k "
            "LDS per wave, VALU busy 0.60, 5.8 waves per SIMD",
}}
BM = "scan_bm_kernel<16, 4, true>"
if BM in vals:
    b = vals[BM]
    per_step = 4  # launches of the kernel per search: the rounds (nearest bucket, next six, the rest) + the retry round
    cyc = b["GRBM_GUI_ACTIVE"] / 8.0 * per_step
    bound["c5_bm"] = {
        "kernel": BM, "rows": 1000000000, "queries": 10000,
        "what": "the bucket-major rounds of the 1B x 16 B, 10 k-query step (scale_base): sums over the %d launches of a step" % per_step,
        "bound": "the memory side: a bucket of 1B rows is four times an XCD's L2, so most groups of a bucket take its rows "
                 "from the Infinity Cache / HBM (L2 hit rate below); memory_side_bytes_per_step over the step's kernel time "
                 "(~66 ms: profiles/r03_default_kernel_stats.csv) is ~6.6 TB/s, the fabric's ceiling.  Instruction issue no "
                 "longer binds (valu_busy_frac; it did before the last kernel changes: 0.69)",
        "kernel_cycles_per_step": round(cyc),
        "valu_busy_frac": round(b["SQ_ACTIVE_INST_VALU"] * per_step * 4 / (n_simd * cyc), 3),
        "valu_insts_per_wave_step": round(b["SQ_INSTS_VALU"] / b["SQ_INSTS_VMEM_RD"], 1),
        "salu_insts_per_wave_step": round(b["SQ_INSTS_SALU"] / b["SQ_INSTS_VMEM_RD"], 1),
        "lds_insts_per_wave_step": round(b["SQ_INSTS_LDS"] / b["SQ_INSTS_VMEM_RD"], 1),
        "wave_steps_per_step": round(b["SQ_INSTS_VMEM_RD"] * per_step),
        "lds_bank_conflict_share": round(b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"], 3),
        "wave_wait_share": round(b["SQ_WAIT_ANY"] / b["SQ_WAVE_CYCLES"], 3),
        "mean_waves_per_simd": round(b["SQ_WAVE_CYCLES"] * 4 / (n_simd * b["GRBM_GUI_ACTIVE"] / 8.0), 2),
        "l2_hit_rate": round(b["TCC_HIT_sum"] / (b["TCC_HIT_sum"] + b["TCC_MISS_sum"]), 3) if "TCC_HIT_sum" in b else None,
        "memory_side_bytes_per_step": (b["FETCH_SIZE"] * 2 + b["WRITE_SIZE"]) * 1024 * per_step,
        "database_bytes": 16e9,
        "note": "FETCH_SIZE doubled per the gfx950 rule for 16-B/lane loads; round 2's one-workgroup-per-query form moved "
                "5.04 TB per step for the same job",
    }
bound["lib_source_hash"] = HASH
json.dump(bound, open(os.path.join(dst, R + "_bound.json"), "w"), indent=1)
s = vals[STREAM]
traffic = {"c5_stream": {
    "kernel": STREAM, "rows": 1000000000,
    "fetch_size_kib": s["FETCH_SIZE"], "write_size_kib": s["WRITE_SIZE"],
    "hbm_bytes_per_launch": (s["FETCH_SIZE"] * 2 + s["WRITE_SIZE"]) * 1024,
    "algorithmic_bytes_per_launch": 16e9,
    "source": "profiles/r03_default_pmc_scan.txt (FETCH_SIZE and WRITE_SIZE in separate --pmc passes of the "
              "default bench command)",
    "note": "FETCH_SIZE doubled per the gfx950 rule for 16-B/lane streaming loads; traffic = 1.00 x algorithmic "
            "bytes: every code byte crosses the memory fabric once"}}
traffic["lib_source_hash"] = HASH
json.dump(traffic, open(os.path.join(dst, R + "_traffic.json"), "w"), indent=1)
print(json.dumps(bound, indent=1))
print(json.dumps(traffic, indent=1))
