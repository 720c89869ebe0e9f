#!/bin/bash
# round 3, call E: full GPU suite + default bench line on the tree with F(4x4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03e
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt; grep -E "F\(4x4,3x3\) forced|resnet train golden" $O/pytest_gpu.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03e/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "executed", d["roofline"]["executed_frac"], "split", d.get("split_precision",{}).get("value"))
for s in d.get("secondary",[]): print(s["config"], s["value"], s["ms_per_step"], s["roofline"]["executed_frac"])
PY
