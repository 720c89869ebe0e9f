#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_train2" -o train -- python "$R/bench.py" --mode train --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/rocprof_train2.log" 2>&1); echo "rc=$?"
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
(cd /tmp && timeout 600 rocprofv3 --pmc $C1 --kernel-trace -d "$R/gpurun_out/pmc_train" -o p -- python "$R/bench.py" --mode train --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmc_train.log" 2>&1); echo "rc=$?"
