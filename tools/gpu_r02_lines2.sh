#!/bin/bash
# vgg path after the upsample convs moved to the Winograd transposed-conv kernels: parity subset + the two vgg_q lines
O=gpurun_out/r02final
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vgg or golden or structured or hourglass or train_steps or replication or determin or ragged or data_parallel" 2>&1 | tail -3
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log | cut -c1-200
