#!/bin/bash
mkdir -p gpurun_out/r02m
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wgrad_winograd or train_steps_golden or hourglass_variants or replication or vgg_f_train or determin" 2>&1 | tail -3
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > gpurun_out/r02m/bench_$n.log 2>&1; tail -1 gpurun_out/r02m/bench_$n.log | cut -c1-170; }
line train --mode train --steps 4 --warmup 2
