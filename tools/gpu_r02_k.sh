#!/bin/bash
mkdir -p gpurun_out/r02k
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv1x1 or resnet_h_train or resnet_f_train or resnet_training" 2>&1 | tail -4
timeout 300 python tools/microbench_conv1x1.py --batch 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r02k/microbench_conv1x1_b16.txt; cat gpurun_out/r02k/microbench_conv1x1_b16.txt | cut -c100-260
timeout 300 python tools/microbench_conv1x1.py --batch 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r02k/microbench_conv1x1_b128.txt; tail -1 gpurun_out/r02k/microbench_conv1x1_b128.txt
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > gpurun_out/r02k/bench_$n.log 2>&1; tail -1 gpurun_out/r02k/bench_$n.log | cut -c1-170; }
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
line resnet_h_train128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
