#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export DREAM_BENCH_BACKEND=gloo
run() { n=$1; shift; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29540 + RANDOM % 100)) bench.py --gpus 2 "$@" > gpurun_out/n2_$n.log 2>&1; echo "rc=$?"; grep -n "Error" gpurun_out/n2_$n.log | head -5; grep '^{"metric' gpurun_out/n2_$n.log | cut -c1-330; }
run inf --steps 3 --warmup 1 --batch 16 --no-split-leg
run train --steps 2 --warmup 1 --batch 8 --mode train
run rtrain --steps 2 --warmup 1 --batch 4 --arch resnet_h --mode train
timeout 600 python -m pytest tests -m gpu -q -k "train_step or deterministic" 2>&1 | tail -1
