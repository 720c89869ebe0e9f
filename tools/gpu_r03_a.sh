#!/bin/bash
# round 3, call A: data-parallel GPU tests, default bench line (secondary block), self-launched 2-rank rehearsal on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "data_parallel or allreduce" -s > $O/pytest_dp.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_dp.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 3000 $O/bench_default.json
DREAM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-split-leg > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "bench2 rc=$?"
tail -c 2500 $O/bench_2rank_gloo.json; tail -5 $O/bench_2rank_gloo.err
