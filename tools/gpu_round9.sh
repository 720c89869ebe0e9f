#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== A/B f16x3 arms at B=128"; timeout 900 python tools/ab_variants.py --f16x3 > gpurun_out/ab_f16x3.log 2>&1; tail -12 gpurun_out/ab_f16x3.log
echo "== pytest gpu (all)"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
