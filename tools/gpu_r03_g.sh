#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03g
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wgrad or training_is_deterministic or train_step" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python bench.py --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rh16.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_rh16.json').read().strip().splitlines()[-1]); print('resnet_h train16:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms')"
timeout 300 python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_train.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_train.json').read().strip().splitlines()[-1]); print('vgg_q train128:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms')"
export TMPDIR=/tmp; R="$PWD"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > "$R/$O/rocprof_rtrain.log" 2>&1); echo "rocprof rc=$?"
db=$(ls $O/prof_rtrain/*.db $O/prof_rtrain/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" $O/bench_resnet_h_train16 > /dev/null 2>&1; rm -rf $O/prof_rtrain; head -30 $O/bench_resnet_h_train16_kernel_stats.csv | cut -c1-150
