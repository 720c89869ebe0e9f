#!/bin/bash
# first-conv rewrite + F(4x4) pool fusion: parity subset, headline bench, kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "first or vgg or structured or golden or winograd4 or pool" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('headline:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms'); print(json.dumps(d.get('secondary')))"
export TMPDIR=/tmp; R="$PWD"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$O/prof" -o b -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$R/$O/rocprof.log" 2>&1); echo "rocprof rc=$?"
db=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" $O/bench > /dev/null 2>&1; rm -rf $O/prof; head -24 $O/bench_kernel_stats.csv | cut -c1-150
