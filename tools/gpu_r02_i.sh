#!/bin/bash
mkdir -p gpurun_out/r02i
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "winograd or train or determin or replication" 2>&1 | tail -4
timeout 600 python tools/microbench_wgrad_wino.py --batch 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02i/microbench_wgrad_wino_b128.txt | cut -c1-250
timeout 600 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02i/bench_train.json | cut -c1-300
timeout 600 python bench.py --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02i/bench_resnet_h_train16.json | cut -c1-300
timeout 600 python bench.py --arch vgg_f --mode train --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02i/bench_vgg_f_train32.json | cut -c1-300
