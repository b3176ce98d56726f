#!/bin/bash
# tools/phases_run.sh "<bench args>" ... -- on the GPU box: cycles per wave in each phase of the
# best-first kernel (the VAQ_PHASES variant must have been built:
# VAQ_VARIANT=phases VAQ_EXTRA_FLAGS=-DVAQ_PHASES python -m vaq_amd.build), with the kernel time beside it
cd "$GRAFT_REPO_ROOT"
export VAQHIP_LIB=$PWD/vaq_amd/lib/variants/phases/libvaqhip.so
for args in "$@"; do
  echo "== $args"
  python3 bench.py --steps 3 --warmup 1 --no-c5-leg --no-cpu --no-recall $args 2>gpurun_out/.phases.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['headline_kernel']
print('   kernel %s %.4f ms wg %d' % (h['kernel'], h['kernel_ms'], h['workgroups']))"
  grep VAQ_PHASES gpurun_out/.phases.err | tail -1
done
