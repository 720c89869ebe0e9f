#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03d
mkdir -p $O
timeout 300 python tools/wino4_diag.py run --batch 128 > $O/wino4_diag.txt 2>&1; echo "diag rc=$?"; grep -v "Warning\|amdgpu.ids" $O/wino4_diag.txt | tr '|' '\n'
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd4" > $O/pytest_wino4.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_wino4.txt
