#!/bin/bash
# A/B of library builds on one box: build/lib<X>.so are copied over dream_amd/libdream_hip.so in turn.
#   bash tools/gpu_ab_libs.sh "<bench args>" A B A B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args="$1"; shift
cp dream_amd/libdream_hip.so /tmp/lib_keep.so
for e in "$@"; do echo "-- $e"; cp build/lib$e.so dream_amd/libdream_hip.so; timeout 300 python bench.py $args --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"; done
cp /tmp/lib_keep.so dream_amd/libdream_hip.so
