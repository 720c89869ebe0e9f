#!/bin/bash
# knock-out / variant table of the F(4x4) kernel (incl. weight ring 8), then the lines the ConvT rule moves
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03o
mkdir -p $O
timeout 400 python tools/wino4_diag.py run --batch 128 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/wino4_diag.txt; cut -d"|" -f1-3 $O/wino4_diag.txt
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$n', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms', 'executed_frac', round(d['roofline']['executed_frac'], 3))"; }
line default --steps 10 --warmup 3
line vgg_f_b32 --arch vgg_f --batch 32 --steps 10 --warmup 3
line resnet_f_b32 --arch resnet_f --batch 32 --steps 10 --warmup 3
