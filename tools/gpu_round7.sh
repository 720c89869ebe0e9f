#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== bench default (fp32 headline + split leg + cpu baseline)"; timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
echo "== rocprof fp16x3"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_f16x3" -o f16 -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --precision fp16x3 > "$R/gpurun_out/rocprof_f16x3.log" 2>&1); echo "rc=$?"
