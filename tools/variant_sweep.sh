#!/bin/bash
# tools/variant_sweep.sh "<bench args>" variant... -- the same bench line for the main library
# and for each experiment library under vaq_amd/lib/variants/ (see vaq_amd/build.py)
cd "$GRAFT_REPO_ROOT"
args=$1; shift
echo "main:"; bash tools/quick_bench.sh "$args"
for v in "$@"; do
  echo "$v:"; VAQHIP_LIB=$PWD/vaq_amd/lib/variants/$v/libvaqhip.so bash tools/quick_bench.sh "$args"
done
