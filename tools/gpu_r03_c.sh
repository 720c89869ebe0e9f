#!/bin/bash
# round 3, call C: knock-out timing + PMC passes of the F(4x4) kernel; resnet goldens again
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03c
mkdir -p $O
timeout 300 python tools/wino4_diag.py run --batch 128 > $O/wino4_diag.txt 2>&1; echo "diag rc=$?"; grep -v Warning $O/wino4_diag.txt
bash tools/gpu_pmc_one.sh wino4 0 100 256 256 128 3 > $O/pmc_one.log 2>&1; cp gpurun_out/pmc_one.txt $O/pmc_wino4_100_256.txt; cat $O/pmc_wino4_100_256.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reference_golden" -s > $O/pytest_goldens.txt 2>&1; echo "pytest goldens rc=$?"
grep -E "resnet train golden|passed|failed|Error|assert" $O/pytest_goldens.txt | tail -12
