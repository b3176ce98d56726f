#!/bin/bash
# tools/quick_bench.sh "<bench args>" ... -- one summary line per argument set (C2 unless told otherwise)
cd "$GRAFT_REPO_ROOT"
for args in "$@"; do
  python3 bench.py --steps 20 --warmup 5 --no-c5-leg --no-cpu --no-recall $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['headline_kernel']; o=h.get('other_kernels_ms',{})
print('%-40s qps %.3fM step %.4f ms kernel %s %.4f ms lds %d wg %d | project %.4f lut %.4f pre-pass %.4f' % ('$args', d['value']/1e6, d['ms_per_step'], h['kernel'], h['kernel_ms'], h['lds_bytes'], h['workgroups'], o.get('project',0), o.get('lut_build',0), o.get('threshold_seed',0)))"
done
