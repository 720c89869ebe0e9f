#!/bin/bash
# persistent Winograd kernel: parity subset, per-layer microbench, headline + training lines
mkdir -p gpurun_out/r02h
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "winograd or golden or structured or train_step or hourglass_variants or ragged" 2>&1 | tail -5
timeout 600 python tools/microbench_wino.py --batch 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02h/microbench_wino_b128.txt | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --no-split-leg 2>&1 | tail -1 | tee gpurun_out/r02h/bench_default.json | cut -c1-300
timeout 600 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02h/bench_train.json | cut -c1-300
