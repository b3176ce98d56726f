#!/bin/bash
# tools/profile_bm.sh <tag> <rows> [pmc] [extra bm_bench args] -- on the GPU box (via gpurun): tools/bm_bench.py
# (bucket-major form only, --skip-base) under rocprofv3:
#   1. --kernel-trace --stats -> gpurun_out/prof_<tag>/kernel_stats.csv
#   2. with `pmc`: FETCH_SIZE / WRITE_SIZE / SQ counters of the scan kernels, each in its own pass
# The library is prebuilt (VAQHIP_LIB names it: build_lib() then spawns nothing under the profiler).
set -o pipefail
tag=$1; rows=$2; shift; shift
pmc=0; [ "$1" = "pmc" ] && { pmc=1; shift; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m vaq_amd.build > /dev/null || exit 1
export VAQ_NO_BUILD=1
out=gpurun_out/prof_$tag
mkdir -p $out
ARGS="--rows $rows --skip-base --steps 3 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/bm_bench.py $ARGS > $out/kt_bench.txt 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 - "$(find $out/kt -name '*kernel_trace.csv' | head -1)" > $out/scan_dispatches.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "scan_b" in n or "bm_" in n:
        print(n.split("(")[0].replace("void vaq::", "").replace("vaq::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms")
PY
rm -rf $out/kt
echo "kernel trace done"; grep -E "scan_|bm_|merge|lut_|project|cost|Name" $out/kernel_stats.csv | cut -c1-200 | head -20
tail -3 $out/kt_bench.txt
[ $pmc = 1 ] || exit 0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-include-regex "scan_" --output-format csv -d $out/pmc_$name -- python3 tools/bm_bench.py $ARGS > /dev/null 2> $out/pmc_$name.err || { echo "pmc pass failed: $pass"; tail -3 $out/pmc_$name.err; continue; }
  f=$(find $out/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f >> $out/pmc_summary.txt
  rm -rf $out/pmc_$name
  echo "pmc pass done: $name"
done
cat $out/pmc_summary.txt
