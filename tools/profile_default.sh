#!/bin/bash
# tools/profile_default.sh <tag> [pmc] -- on the GPU box (via gpurun): the DEFAULT bench command
# (`python3 bench.py --steps 20 --warmup 5`, what the driver runs) under rocprofv3:
#   1. --kernel-trace --stats  -> gpurun_out/prof_<tag>/kernel_stats.csv + the bench line
#   2. with `pmc`: counters of the scan kernels in their own runs (never combined with trace
#      domains; FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots)
# Copy what is to be judged into profiles/ (tools/profile_collect.py does it).
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# build BEFORE the profiler is in the process tree (hipcc / make children would be the exec-after-GPU-init
# hop this pool forbids); the profiled commands then only check that the libraries exist
python3 -m vaq_amd.build > /dev/null && make -s -C oracle all || exit 1
export VAQ_NO_BUILD=1
out=gpurun_out/prof_$tag
mkdir -p $out
BARGS="--steps 20 --warmup 5 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py $BARGS > $out/kt_bench.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/kt  # (the raw trace of a 1B-row build is large; gpurun returns at most 64 MiB)
echo "kernel trace done"; head -4 $out/kernel_stats.csv | cut -c1-160
[ "$1" = "pmc" ] || exit 0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-include-regex "scan_" --output-format csv -d $out/pmc_$name -- python3 bench.py $BARGS --no-cpu --no-recall > /dev/null 2> $out/pmc_$name.err || { echo "pmc pass failed: $pass"; tail -3 $out/pmc_$name.err; continue; }
  f=$(find $out/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f >> $out/pmc_summary.txt
  rm -rf $out/pmc_$name
  echo "pmc pass done: $name"
done
cat $out/pmc_summary.txt
