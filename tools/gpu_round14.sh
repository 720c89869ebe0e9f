#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { echo "-- $1"; env $1 timeout 300 python bench.py --arch $2 --mode train --batch $3 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for e in "X=1" "DREAM_WGRAD_NPMAX=187" "DREAM_WGRAD_WGS_WIDE=512" "DREAM_WGRAD_WGS_WIDE=512 DREAM_WGRAD_NPMAX=187" "DREAM_WGRAD_WGS_WIDE=512 DREAM_WGRAD_NPMAX=187 DREAM_WGRAD_WGS=512" "DREAM_WGRAD_WGS_WIDE=256 DREAM_WGRAD_NPMAX=187"; do run "$e" resnet_h 16; done > gpurun_out/ab_wgrad.log 2>&1
for e in "X=1" "DREAM_WGRAD_NPMAX=187" "DREAM_WGRAD_WGS=512"; do run "$e" vgg_q 64; done >> gpurun_out/ab_wgrad.log 2>&1
cat gpurun_out/ab_wgrad.log
