#!/bin/bash
# PMC passes on one layer / one kernel:  bash tools/gpu_pmc_one.sh <one_layer.py args...>   (output: gpurun_out/pmc_one.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pmc_one
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"
P3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
P4="TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
for pass in 1 2 3 4; do
  eval C=\$P$pass
  (cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/p$pass" -o p -- python "$R/tools/one_layer.py" "$@" > "$R/$O/p$pass.log" 2>&1); echo "pass $pass rc=$?"
done
python - <<'PY' | tee gpurun_out/pmc_one.txt
import sqlite3, glob
for d in sorted(glob.glob("gpurun_out/pmc_one/p*/")):
    for f in glob.glob(d + "**/*.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%kernel%' and kernel_name not like '%reduce%' and kernel_name not like '%at::%' group by kernel_name, counter_name"))
        for r in rows: print("   %-44s %-34s %16.0f (n=%d)" % (r[0][-44:], r[1], r[2], r[3]))
PY
rm -rf $O
