#!/bin/bash
# One gpurun call: smoke, GPU parity tests, per-layer microbenchmark, bench line, rocprof kernel stats.
# Everything is logged under gpurun_out/ (merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.log
nproc >> gpurun_out/device.log; lscpu | grep "Model name" >> gpurun_out/device.log
echo "== smoke" ; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== microbench"; timeout 600 python tools/microbench.py --batch ${MB_BATCH:-32} > gpurun_out/microbench.log 2>&1; echo "microbench rc=$?"; tail -22 gpurun_out/microbench.log
echo "== bench"; timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r01 -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"; tail -3 gpurun_out/rocprof.log
ls -R gpurun_out/prof 2>/dev/null | head -20
