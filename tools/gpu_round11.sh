#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (variants)"; timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -k "hourglass_variants" > gpurun_out/pytest_gpu_variants.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu_variants.log
