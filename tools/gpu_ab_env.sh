#!/bin/bash
# A/B of environment switches on one box: bash tools/gpu_ab_env.sh "<bench args>" "ENV=a" "ENV=b" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args="$1"; shift
for rep in 1 2; do for e in "$@"; do echo "-- $e"; env $e timeout 300 python bench.py $args --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; done; done
