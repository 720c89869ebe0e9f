#!/bin/bash
# GPU parity subset for the ResNet path + its micro-benchmarks and bench lines (collected like tools/gpu_r02_final.sh's)
O=gpurun_out/r02final
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resnet or conv_transpose4x4 or conv4x4s2" 2>&1 | tail -3
for b in 32 128 16; do timeout 300 python tools/microbench_convT.py --batch $b 2>&1 | grep -v amdgpu.ids; done > $O/microbench_convT.txt; grep "sum over" $O/microbench_convT.txt
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | cut -c1-160; }
line resnet_f_b32 --arch resnet_f --batch 32
line resnet_h_b128 --arch resnet_h --batch 128
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 5 --warmup 2
line resnet_h_train128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
