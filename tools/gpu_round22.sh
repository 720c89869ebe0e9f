#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for a in vgg_q resnet_h resnet_f; do for b in 1 8; do timeout 300 python bench.py --arch $a --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-split-leg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', $b, round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms', 'conv share', round(d['roofline']['share_of_step_time'],3), 'launches/step', d['roofline']['launches']/50)"; done; done 2>&1 | tee gpurun_out/latency.log
