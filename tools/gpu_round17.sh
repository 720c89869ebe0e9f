#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python tools/microbench.py --batch 64 --variants=-1 > gpurun_out/microbench_f16_b.log 2>&1; tail -30 gpurun_out/microbench_f16_b.log
