#!/bin/bash
# tools/quick_bench.sh "<bench args>" ... -- one summary line per argument set (C2 unless told otherwise)
cd "$GRAFT_REPO_ROOT"
for args in "$@"; do
  python3 bench.py --steps 20 --warmup 5 --no-c5-leg --no-cpu --no-recall $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['headline_kernel']
print('%-40s qps %.3fM step %.4f ms kernel %s %.4f ms lds %d wg %d' % ('$args', d['value']/1e6, d['ms_per_step'], h['kernel'], h['kernel_ms'], h['lds_bytes'], h['workgroups']))"
done
