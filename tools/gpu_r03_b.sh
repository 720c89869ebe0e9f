#!/bin/bash
# round 3, call B: F(4x4,3x3) kernel -- parity, microbench vs F(2x2), resnet training goldens, structured fixtures
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd4" -s > $O/pytest_wino4.txt 2>&1; echo "pytest wino4 rc=$?"
grep -E "rel err|passed|failed|Error" $O/pytest_wino4.txt | tail -25
timeout 600 python tools/microbench_wino4.py --batch 128 --reps 5 --out $O/microbench_wino4_b128.json > $O/microbench_wino4_b128.txt 2>&1; echo "microbench rc=$?"
cat $O/microbench_wino4_b128.txt | grep -v Warning
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reference_golden or structured" -s > $O/pytest_goldens.txt 2>&1; echo "pytest goldens rc=$?"
grep -E "resnet train golden|structured|passed|failed|Error|assert" $O/pytest_goldens.txt | tail -30
