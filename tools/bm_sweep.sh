for nq in 128 512 2048; do
  echo "== 125M x16B nq=$nq"; timeout -k 10 300 python tools/bm_bench.py --rows 125000000 --nq $nq --steps 3 --rounds 1:6,0:6 2>&1 | grep label | cut -c1-200
done
echo "== C4 100M x 8B nq=10000"; timeout -k 10 300 python tools/bm_bench.py --rows 100000000 --m 8 --nq 10000 --steps 3 --rounds 1:6,1:3,0:6 2>&1 | grep label | cut -c1-200
echo "== C4 100M x 8B nq=1000"; timeout -k 10 300 python tools/bm_bench.py --rows 100000000 --m 8 --nq 1000 --steps 3 --rounds 1:6 2>&1 | grep label | cut -c1-200
