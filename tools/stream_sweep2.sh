#!/bin/bash
# tools/stream_sweep2.sh variant... -- the streaming pass (250M x 16 B, every bucket visited), default launch shape only
cd "$GRAFT_REPO_ROOT"
run() {
  python3 bench.py --workload c5 --rows 250000000 --random-codes --nq 2 --no-skip --seed 0 --steps 20 --warmup 3 --no-cpu --no-recall $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['headline_kernel']; r=d['roofline']
print('%-28s kernel %s %.4f ms  %.0f GB/s frac %.3f  wg %d lds %d' % ('$1', h['kernel'], h['kernel_ms'], r.get('achieved') or 0, r.get('frac') or 0, h['workgroups'], h['lds_bytes']))"
}
echo main; run ""
for v in "$@"; do echo $v; export VAQHIP_LIB=$PWD/vaq_amd/lib/variants/$v/libvaqhip.so; run ""; done
