#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/rehearsal
run() { name=$1; shift; HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 5 70 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --one-device --no-cpu "$@" > gpurun_out/rehearsal/b_$name.json 2> gpurun_out/rehearsal/b_$name.err; echo "$name rc=$?" | tee -a gpurun_out/rehearsal/bisect.txt; }
rm -f gpurun_out/rehearsal/bisect.txt
run small --rows 8000000 --nq 1024
run bf0 --rows 64000000 --nq 1024 --bf 0
run nq64 --rows 64000000 --nq 64
run dflt --rows 64000000 --nq 1024
