#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== A/B variants at B=128"; timeout 900 python tools/ab_variants.py > gpurun_out/ab_variants.log 2>&1; tail -12 gpurun_out/ab_variants.log
echo "== pytest gpu (pool fusion, inference)"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -k "fused_maxpool or inference_golden or full_size" > gpurun_out/pytest_gpu_pool.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_pool.log
echo "== bench default"; sleep 15; timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_default2.log 2>&1; tail -1 gpurun_out/bench_default2.log
