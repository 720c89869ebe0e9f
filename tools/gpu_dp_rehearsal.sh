#!/bin/bash
# Rehearsal of the N > 1 bench path on a single-GPU box: two ranks share device 0 over gloo (RCCL refuses two ranks on
# one device).  Checks that the multi-process logic runs end to end -- barrier/max timing, data-parallel gradient
# exchange -- not performance: two processes time-slice one GPU.  (With DREAM_OVERLAP_WGRAD=1 and two processes on one
# GPU the side stream makes the time-slicing pathological, 2.8 s/step; a dedicated GPU per rank does not have this.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export DREAM_BENCH_BACKEND=gloo DREAM_OVERLAP_WGRAD=0
run() { n=$1; shift; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29540 + RANDOM % 100)) bench.py --gpus 2 "$@" > gpurun_out/n2_$n.log 2>&1; echo "$n rc=$?"; grep -n "Error" gpurun_out/n2_$n.log | head -5; grep '^{"metric' gpurun_out/n2_$n.log | cut -c1-230; }
run inf --steps 3 --warmup 1 --batch 16 --no-split-leg
run train --steps 2 --warmup 1 --batch 8 --mode train
run rtrain --steps 2 --warmup 1 --batch 4 --arch resnet_h --mode train
# the gradient exchange with the weight-gradient side stream, one rank (gloo world of 1)
unset DREAM_BENCH_BACKEND
DREAM_OVERLAP_WGRAD=1 DREAM_FORCE_REDUCER=1 timeout 300 python bench.py --arch resnet_h --mode train --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230
