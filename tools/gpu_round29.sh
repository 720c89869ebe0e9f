#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
timeout 600 python -m pytest tests -m gpu -q -k "peak or full_size or inference_golden" 2>&1 | tail -1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_rf" -o rf -- python "$R/bench.py" --arch resnet_f --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-split-leg > "$R/gpurun_out/rocprof_rf.log" 2>&1); echo "rc=$?"; grep -h '^{"metric' gpurun_out/rocprof_rf.log | cut -c1-200
