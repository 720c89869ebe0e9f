#!/bin/bash
# The full GPU suite + smoke() on the committed tree (the library is unchanged since tools/gpu_r04_final.sh ran; only Python changed).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04tests
mkdir -p $O
echo "== pytest gpu (all)"; timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
