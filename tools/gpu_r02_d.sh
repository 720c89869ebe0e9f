#!/bin/bash
# Round 2, call D: whole GPU suite with Winograd as the default fp32 conv path and the data-parallel executor, default bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log; grep "^FAILED\|^E " $O/pytest_gpu.log | head -20
echo "== default bench"; timeout 600 python bench.py --cpu-seconds 6 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-1500
echo "== bench direct"; timeout 600 python bench.py --no-cpu-baseline --no-split-leg --conv-algorithm direct > $O/bench_direct.log 2>&1; tail -1 $O/bench_direct.log | cut -c1-300
echo "== bench train"; timeout 600 python bench.py --mode train --steps 4 --warmup 1 > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log | cut -c1-600
echo "== bench single-process dp2 on one GPU (functional)"; DREAM_BENCH_GPU_IDS=0,0 timeout 600 python bench.py --gpus 2 --single-process --batch 32 --no-cpu-baseline --no-split-leg > $O/bench_sp2.log 2>&1; tail -1 $O/bench_sp2.log | cut -c1-300
