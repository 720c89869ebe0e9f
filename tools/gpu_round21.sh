#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pmcx
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_$C" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg > "$R/$O/pmc_$C.log" 2>&1); echo "rc=$?"
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db $O/traffic.json
