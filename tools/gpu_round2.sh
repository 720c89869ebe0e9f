#!/bin/bash
# GPU round 2: training bench (configs[2]) + rocprof of it, PMC passes (HBM traffic) for the inference bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== train bench"; timeout 900 python bench.py --mode train --steps 3 --warmup 1 > gpurun_out/bench_train.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench_train.log
echo "== rocprof train"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_train" -o r01t -- python "$R/bench.py" --mode train --steps 1 --warmup 1 > "$R/gpurun_out/rocprof_train.log" 2>&1); echo "rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"; (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$R/gpurun_out/pmc_$C" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmc_$C.log" 2>&1); echo "rc=$?"; tail -2 "$R/gpurun_out/pmc_$C.log"
done
echo "== inference bench (cool)"; sleep 20; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cool.log 2>&1; tail -1 gpurun_out/bench_cool.log
ls gpurun_out gpurun_out/pmc_FETCH_SIZE 2>/dev/null | head -30
