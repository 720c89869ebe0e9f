#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { echo "-- $1"; cp build/lib$1.so dream_amd/libdream_hip.so; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-split-leg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"; }
for e in A G A G; do run "$e"; done 2>&1 | tee gpurun_out/ab_convsched.log
cp build/libA.so dream_amd/libdream_hip.so
