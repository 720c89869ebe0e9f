#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resnet or determin" 2>&1 | tail -3
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > gpurun_out/r02l/bench_$n.log 2>&1; tail -1 gpurun_out/r02l/bench_$n.log | cut -c1-170; }
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
line resnet_h_train128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
timeout 300 python tools/layer_profile.py --arch resnet_h --mode train --batch 16 --top 30 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > gpurun_out/r02l/layer_profile_resnet_h_train16.txt; sed -n 1,30p gpurun_out/r02l/layer_profile_resnet_h_train16.txt | cut -c1-150
