#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
echo "== bench vgg_q"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_vggq.log 2>&1; tail -1 gpurun_out/bench_vggq.log
echo "== bench resnet_h train b=16 (configs[3] per-GPU share)"; timeout 600 python bench.py --arch resnet_h --mode train --batch 16 --steps 5 --warmup 2 > gpurun_out/bench_resnet_h_train16.log 2>&1; tail -1 gpurun_out/bench_resnet_h_train16.log
echo "== bench resnet_h train b=128"; timeout 600 python bench.py --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1 > gpurun_out/bench_resnet_h_train128.log 2>&1; tail -1 gpurun_out/bench_resnet_h_train128.log
echo "== rocprof resnet_h train b=16"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_resnet_h_train" -o rt -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 2 --warmup 1 > "$R/gpurun_out/rocprof_resnet_h_train.log" 2>&1); echo "rc=$?"
echo "== bench vgg_q train"; timeout 600 python bench.py --mode train --steps 3 --warmup 1 > gpurun_out/bench_vggq_train.log 2>&1; tail -1 gpurun_out/bench_vggq_train.log
