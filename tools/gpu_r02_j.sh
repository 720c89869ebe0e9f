#!/bin/bash
mkdir -p gpurun_out/r02j
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv1x1 or resnet or determin or replication or checkpoint" 2>&1 | tail -4
timeout 300 python tools/microbench_conv1x1.py --batch 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r02j/microbench_conv1x1_b16.txt; tail -1 gpurun_out/r02j/microbench_conv1x1_b16.txt
timeout 300 python tools/microbench_conv1x1.py --batch 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r02j/microbench_conv1x1_b128.txt; tail -1 gpurun_out/r02j/microbench_conv1x1_b128.txt
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > gpurun_out/r02j/bench_$n.log 2>&1; tail -1 gpurun_out/r02j/bench_$n.log | cut -c1-170; }
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
DREAM_CONV1X1_ALGORITHM=direct line resnet_h_train16_direct1x1 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
line resnet_h_train128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
line resnet_f_b32 --arch resnet_f --batch 32 --no-split-leg
line resnet_h_b128 --arch resnet_h --batch 128 --no-split-leg
