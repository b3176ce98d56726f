#!/bin/bash
# tools/profile_gpu.sh <tag> [bench args...]
# Runs on the GPU box (via gpurun): kernel-trace stats, then PMC passes in their
# own runs (never combined with trace domains), for the scan kernel.
# Results land in gpurun_out/prof_<tag>/; copy the summaries into profiles/.
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m vaq_amd.build > /dev/null && make -s -C oracle all || exit 1   # (no compiler children under the profiler)
export VAQ_NO_BUILD=1
out=gpurun_out/prof_$tag
mkdir -p $out
BARGS="--steps 5 --warmup 2 --no-cpu --no-recall $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py $BARGS > $out/kt_bench.json 2> $out/kt.err || exit 1
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/kt
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-include-regex "scan_" --output-format csv -d $out/pmc_$name -- python3 bench.py $BARGS > /dev/null 2> $out/pmc_$name.err || { echo "pmc pass failed: $pass"; tail -3 $out/pmc_$name.err; continue; }
  f=$(find $out/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f >> $out/pmc_summary.txt
  rm -rf $out/pmc_$name
done
cat $out/pmc_summary.txt
