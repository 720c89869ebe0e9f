#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { echo "-- $1 $2"; cp build/lib$1.so dream_amd/libdream_hip.so; timeout 300 python bench.py --arch $2 --batch $3 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['split_precision']['value'],1), round(d['split_precision']['roofline']['frac'],4))"; }
for e in Y Z Y Z; do run "$e" resnet_f 32; done 2>&1 | tee gpurun_out/ab_xcd16.log
for e in Y Z; do run "$e" vgg_f 32; done 2>&1 | tee -a gpurun_out/ab_xcd16.log
cp build/libY.so dream_amd/libdream_hip.so
