#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== rocprof fp32 default"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_fp32" -o fp32 -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-split-leg > "$R/gpurun_out/rocprof_fp32.log" 2>&1); echo "rc=$?"; tail -1 gpurun_out/rocprof_fp32.log | cut -c1-400
echo "== rocprof train"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_train" -o train -- python "$R/bench.py" --mode train --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/rocprof_train.log" 2>&1); echo "rc=$?"; tail -1 gpurun_out/rocprof_train.log | cut -c1-400
echo "== rocprof resnet_h train b16"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/rocprof_rtrain.log" 2>&1); echo "rc=$?"; tail -1 gpurun_out/rocprof_rtrain.log | cut -c1-400
