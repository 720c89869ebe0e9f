#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bench() { timeout 600 python bench.py "$@" --no-cpu-baseline --no-split-leg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],3))"; }
timeout 900 python -m pytest tests -m gpu -q -k "resnet or conv" 2>&1 | tail -1
bench --arch resnet_h --mode train --batch 16 --steps 5 --warmup 2
bench --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
bench --arch resnet_f --batch 32
bench --arch resnet_h --batch 128
bench --arch resnet_h --batch 1 --steps 50 --warmup 10
bench
