#!/bin/bash
# resnet_f inference b=32 (one GPU's share of configs[4]): kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp; R="$PWD"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$O/prof" -o b -- python "$R/bench.py" --arch resnet_f --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$R/$O/rocprof.log" 2>&1); echo "rocprof rc=$?"
db=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" $O/bench_resnet_f_b32 > /dev/null 2>&1; rm -rf $O/prof
python - <<'PY'
import csv, re
rows=list(csv.DictReader(open('gpurun_out/r03m/bench_resnet_f_b32_kernel_stats.csv')))
tot=sum(float(r['total_ms']) for r in rows if 'family' not in r['kernel'])
print('total ms', tot)
for r in rows[:28]:
    name=re.sub(r'\(anonymous namespace\)::','',r['kernel']); name=re.sub(r'^void ','',name).split('(')[0][:52]
    print('%-54s %5s %8.2f ms %5.1f%% avg %7.1f us'%(name, r['calls'], float(r['total_ms']), 100*float(r['total_ms'])/tot, float(r['avg_us'])))
PY
