#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
i=0
for cfg in "f16x3 1 100 256 256 32" "f16x3 4 100 256 256 32" "fp32 3 100 256 256 32" "f16x3 1 400 64 64 32"; do
  i=$((i+1))
  for pass in 1 2; do
    if [ $pass = 1 ]; then C="$C1"; else C="$C2"; fi
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$R/gpurun_out/pmc_layer_${i}_${pass}" -o p -- python "$R/tools/one_layer.py" $cfg > "$R/gpurun_out/pmc_layer_${i}_${pass}.log" 2>&1); echo "cfg $i pass $pass rc=$?"
  done
done
python - <<'PY'
import sqlite3, glob, os
for d in sorted(glob.glob("gpurun_out/pmc_layer_*_*/")):
    for f in glob.glob(d + "*.db"):
        cur = sqlite3.connect(f).cursor()
        try:
            rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%conv%kernel%' group by kernel_name, counter_name"))
        except Exception as e:
            print(d, "ERR", e); continue
        print(d)
        for r in rows: print("   %-40s %-28s %14.0f (n=%d)" % (r[0][:40], r[1], r[2], r[3]))
PY
