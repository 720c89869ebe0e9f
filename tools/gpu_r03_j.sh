#!/bin/bash
# full GPU suite + training kernel profiles (vgg_q b=128, resnet_h b=16) on the current tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03j
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
export TMPDIR=/tmp; R="$PWD"
for spec in "vgg_q 128 vtrain" "resnet_h 16 rtrain"; do
  set -- $spec
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_$3" -o $3 -- python "$R/bench.py" --arch $1 --mode train --batch $2 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$R/$O/rocprof_$3.log" 2>&1); echo "rocprof $3 rc=$?"
  tail -1 $O/rocprof_$3.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 train$2 (under rocprof):', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 2), 'ms')
except Exception as e: print('no json', e)"
  db=$(ls $O/prof_$3/*.db $O/prof_$3/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" $O/bench_$1_train$2 > /dev/null 2>&1; rm -rf $O/prof_$3
  head -28 $O/bench_$1_train$2_kernel_stats.csv | cut -c1-160
done
