cd "$GRAFT_REPO_ROOT"
echo "== default (4096 sample rows)"; timeout -k 10 300 python tools/bm_bench.py --rows 1000000000 --nq 10000 --steps 2 --skip-base --rounds 1:6,1:2,1:12 2>&1 | grep label | cut -c1-170
echo "== 16384 sample rows"; VAQHIP_LIB=$PWD/vaq_amd/lib/variants/boot16k/libvaqhip.so timeout -k 10 300 python tools/bm_bench.py --rows 1000000000 --nq 10000 --steps 2 --skip-base --rounds 1:6 2>&1 | grep label | cut -c1-170
