#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { echo "-- $1"; env $1 timeout 300 python bench.py --mode train --batch $2 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for e in "DREAM_WGRAD_STAGGER=0" "DREAM_WGRAD_STAGGER=2" "DREAM_WGRAD_STAGGER=3" "DREAM_WGRAD_STAGGER=5" "DREAM_WGRAD_STAGGER=0"; do run "$e" 128; done 2>&1 | tee gpurun_out/ab_stagger.log
