#!/bin/bash
# narrow F(4x4) workgroup shape: parity, per-layer microbench, headline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd4" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python tools/microbench_wino4.py --batch 128 --reps 5 --out $O/microbench_wino4_b128.json > $O/microbench_wino4_b128.txt 2>&1; cat $O/microbench_wino4_b128.txt | cut -c1-220
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('headline:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms')"
