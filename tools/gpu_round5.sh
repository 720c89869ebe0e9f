#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu (split precision + resnet ops)"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "split_precision or resnet_training_ops or conv_variants" > gpurun_out/pytest_gpu_f16.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu_f16.log
echo "== microbench f16x3"; timeout 900 python tools/microbench.py --batch 32 --variants=-1 > gpurun_out/microbench_f16.log 2>&1; tail -20 gpurun_out/microbench_f16.log
echo "== bench vgg_q fp32"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_vggq.log 2>&1; tail -1 gpurun_out/bench_vggq.log
echo "== bench vgg_q fp16x3"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision fp16x3 > gpurun_out/bench_vggq_f16x3.log 2>&1; tail -1 gpurun_out/bench_vggq_f16x3.log
echo "== rocprof fp16x3"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_f16x3" -o f16 -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --precision fp16x3 > "$R/gpurun_out/rocprof_f16x3.log" 2>&1); echo "rc=$?"
echo "== bench resnet_h train b=16"; timeout 600 python bench.py --arch resnet_h --mode train --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_resnet_h_train16.log 2>&1; tail -1 gpurun_out/bench_resnet_h_train16.log
