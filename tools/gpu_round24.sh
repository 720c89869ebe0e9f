#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for b in 16 32; do echo "== B=$b"; timeout 600 python tools/microbench_resnet.py --batch $b; done 2>&1 | tee gpurun_out/microbench_resnet.log
