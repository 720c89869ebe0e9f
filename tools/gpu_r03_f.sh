#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03f
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pack_graph or training_is_deterministic or resnet_h_train_step" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
for f in 1 0; do DREAM_PACK_GRAPH=$f timeout 300 python bench.py --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rh16_pg$f.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_rh16_pg$f.json').read().strip().splitlines()[-1]); print('pack graph $f:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms')"; done
