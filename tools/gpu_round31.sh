#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { echo "-- $1 $2 b=$3"; env $1 timeout 300 python bench.py --arch $2 --mode train --batch $3 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for b in 32 64; do for e in "DREAM_OVERLAP_WGRAD=0" "DREAM_OVERLAP_WGRAD=1"; do run "$e" resnet_h $b; done; done
