#!/bin/bash
# Round 2, call B: Winograd kernel on the device (parity + per-layer A/B against the direct kernel) and the reworked
# full-size training tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02b
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest (winograd, full-size training)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -s -k "winograd or full_size_training or first_conv_wgrad" > $O/pytest.log 2>&1; echo "rc=$?"; grep -h "rel err\|vs b=2\|worst\|passed\|failed\|^E " $O/pytest.log | head -40
for V in 0 4 8; do
echo "== microbench b=128 variant $V"; timeout 600 python tools/microbench_wino.py --batch 128 --variant $V --out $O/microbench_wino_b128_v$V.json 2>&1 | tee $O/microbench_wino_b128_v$V.txt
done
echo "== microbench b=16"; timeout 300 python tools/microbench_wino.py --batch 16 --out $O/microbench_wino_b16.json 2>&1 | tee $O/microbench_wino_b16.txt
