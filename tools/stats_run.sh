#!/bin/bash
# tools/stats_run.sh -- on the GPU box: per-wave event counts / cycle shares of the scan kernels
# (the VAQ_STATS variant library must have been built: VAQ_VARIANT=stats VAQ_EXTRA_FLAGS=-DVAQ_STATS)
cd "$GRAFT_REPO_ROOT"
export VAQHIP_LIB=$PWD/vaq_amd/lib/variants/stats/libvaqhip.so
mkdir -p gpurun_out/stats
for args in "$@"; do
  echo "== $args"
  python3 bench.py --steps 2 --warmup 1 --no-c5-leg --no-cpu --no-recall $args 2>&1 >/dev/null | grep VAQ_STATS | tail -1
done
