#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { echo "-- $1"; env $1 timeout 300 python bench.py --arch resnet_h --mode train --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
run "DREAM_OVERLAP_WGRAD=1"
run "DREAM_OVERLAP_WGRAD=1 DREAM_FORCE_REDUCER=1"
run "DREAM_OVERLAP_WGRAD=0 DREAM_FORCE_REDUCER=1"
