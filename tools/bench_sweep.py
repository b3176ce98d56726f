#!/usr/bin/env python3
"""Run bench.py once per argument set and print one summary line each.

    python tools/bench_sweep.py "--workload c3 --bucket-bits 8" "--workload c3 --bucket-bits 10" ...
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + spec.split(), capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        print(spec, "FAILED", r.stderr[-400:], flush=True)
        continue
    rf = d["roofline"]
    print(f"{spec:60s} qps={d['value']:.0f} step={d['ms_per_step']:.3f}ms scan={rf['kernel_ms']:.3f}ms "
          f"other={rf['other_kernels_ms']} lds={rf['lds_bytes']} slices={rf['slices']} "
          f"recall={(d.get('recall') or {}).get('recall_at_100')}", flush=True)
