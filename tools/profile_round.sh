#!/bin/bash
# tools/profile_round.sh <round tag, e.g. r03> -- on the GPU box (via gpurun): everything profiles/ keeps for a
# round, from ONE build: the default bench command under rocprofv3 (kernel trace + PMC passes), the C4 and 125M
# bucket-major runs with their own counters, and the plain (unprofiled) bench lines of the default, C3 and C4
# workloads.  Afterwards, here: python tools/profile_collect.py <tag>; copy gpurun_out/prof_<tag>c4 etc.
tag=$1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
set -o pipefail
echo "== default bench under rocprofv3"; timeout -k 10 900 bash tools/profile_default.sh $tag pmc > gpurun_out/prof_${tag}.log 2>&1 || { tail -5 gpurun_out/prof_${tag}.log; exit 1; }
echo "== c4 under rocprofv3";            timeout -k 10 400 bash tools/profile_gpu.sh ${tag}c4 --workload c4 > gpurun_out/prof_${tag}c4.log 2>&1 || { tail -5 gpurun_out/prof_${tag}c4.log; exit 1; }
echo "== 125M bucket-major under rocprofv3"; timeout -k 10 400 bash tools/profile_bm.sh ${tag}bm125 125000000 pmc > gpurun_out/prof_${tag}bm125.log 2>&1 || { tail -5 gpurun_out/prof_${tag}bm125.log; exit 1; }
echo "== plain bench lines"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_default_bench.json 2> gpurun_out/${tag}_default_bench.err || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --workload c3 --no-c5-leg > gpurun_out/${tag}_c3_bench.json 2>/dev/null || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --workload c4 --no-c5-leg > gpurun_out/${tag}_c4_bench.json 2>/dev/null || exit 1
echo done
