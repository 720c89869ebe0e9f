#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu (training)"; timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -k "train or backward or wgrad or variants" > gpurun_out/pytest_gpu_train.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_train.log
echo "== rocprof resnet_h train b16"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/rocprof_rtrain.log" 2>&1); echo "rc=$?"; grep -h '^{"metric' gpurun_out/rocprof_rtrain.log | cut -c1-330
echo "== bench resnet_h train b128"; timeout 600 python bench.py --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_rtrain128.log 2>&1; tail -1 gpurun_out/bench_rtrain128.log | cut -c1-330
echo "== bench vgg_f train b32"; timeout 600 python bench.py --arch vgg_f --mode train --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_vggf_train.log 2>&1; tail -1 gpurun_out/bench_vggf_train.log | cut -c1-330
