#!/bin/bash
# GPU round 3: regression (tests + vgg_q bench after the kernel generalisation) and the ResNet path.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
echo "== microbench (heuristic only)"; timeout 600 python tools/microbench.py --batch 32 --variants=-1,0,1,3,4 > gpurun_out/microbench2.log 2>&1; tail -18 gpurun_out/microbench2.log
echo "== bench vgg_q"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_vggq.log 2>&1; tail -1 gpurun_out/bench_vggq.log
echo "== bench resnet_f b=32 (configs[4] per-GPU share)"; timeout 600 python bench.py --arch resnet_f --batch 32 --steps 5 --warmup 2 --cpu-seconds 8 > gpurun_out/bench_resnet_f.log 2>&1; tail -1 gpurun_out/bench_resnet_f.log
echo "== bench resnet_h b=128 inference"; timeout 600 python bench.py --arch resnet_h --batch 128 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_resnet_h.log 2>&1; tail -1 gpurun_out/bench_resnet_h.log
echo "== bench vgg_f b=32 inference"; timeout 600 python bench.py --arch vgg_f --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_vgg_f.log 2>&1; tail -1 gpurun_out/bench_vgg_f.log
echo "== rocprof resnet_f"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_resnet_f" -o rf -- python "$R/bench.py" --arch resnet_f --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/rocprof_resnet_f.log" 2>&1); echo "rc=$?"
