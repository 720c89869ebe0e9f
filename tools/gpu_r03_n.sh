#!/bin/bash
# transposed convs on the F(4x4) kernel (25-position phase patterns): parity, microbench, the lines it moves
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "transpose4x4 or structured or golden or resnet" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python tools/microbench_convT.py 2>&1 | grep -v amdgpu.ids > $O/microbench_convT.txt; cat $O/microbench_convT.txt
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$n', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms', 'executed_frac', round(d['roofline']['executed_frac'], 3))"; }
line default --steps 10 --warmup 3
line resnet_f_b32 --arch resnet_f --batch 32 --steps 10 --warmup 3
line resnet_h_b128 --arch resnet_h --batch 128
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
line vgg_f_b32 --arch vgg_f --batch 32 --steps 10 --warmup 3
