#!/bin/bash
# headline kernel stats on the current tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03l
mkdir -p $O
export TMPDIR=/tmp; R="$PWD"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$O/prof" -o b -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$R/$O/rocprof.log" 2>&1); echo "rocprof rc=$?"
db=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" $O/bench > /dev/null 2>&1; rm -rf $O/prof; head -24 $O/bench_kernel_stats.csv | cut -c1-150
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03l/bench_conv_dispatches.csv')))
# one step's conv dispatches in order (last step)
n=len(rows)//7
for r in rows[-n:]:
    print(r['kernel'][:60], r['duration_us'], r['grid_x'], r['grid_y'], r['lds_bytes'])
PY
