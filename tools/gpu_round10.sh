#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu (all)"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== bench resnet_f B=32 (fp32 + split leg)"; timeout 600 python bench.py --arch resnet_f --batch 32 --no-cpu-baseline > gpurun_out/bench_resnet_f.log 2>&1; tail -1 gpurun_out/bench_resnet_f.log
echo "== bench resnet_h B=128 (fp32 + split leg)"; timeout 600 python bench.py --arch resnet_h --batch 128 --no-cpu-baseline > gpurun_out/bench_resnet_h.log 2>&1; tail -1 gpurun_out/bench_resnet_h.log
echo "== rocprof resnet_f fp16x3"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_resnet_f16" -o rf16 -- python "$R/bench.py" --arch resnet_f --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --precision fp16x3 > "$R/gpurun_out/rocprof_resnet_f16.log" 2>&1); echo "rc=$?"
python tools/prof_summary.py gpurun_out/prof_resnet_f16 2>/dev/null | head -30
