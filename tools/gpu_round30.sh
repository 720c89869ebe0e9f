#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 120 python tools/wgrad_time.py 2>&1 | tail -1
bench() { timeout 600 python bench.py "$@" --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],3))"; }
bench --mode train --steps 4 --warmup 1
bench --arch resnet_h --mode train --batch 16 --steps 5 --warmup 2
bench --arch vgg_f --mode train --batch 32 --steps 3 --warmup 1
timeout 600 python -m pytest tests -m gpu -q -k "train or backward or wgrad" 2>&1 | tail -1
