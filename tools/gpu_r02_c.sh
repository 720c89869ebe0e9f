#!/bin/bash
# Round 2, call C: hardware counters of the Winograd kernel on two layer shapes (SQ issue/wait/MFMA, LDS, L1/L2 traffic).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02c
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
(cd /tmp && rocprofv3 -L > "$R/$O/counters_list.txt" 2>&1); grep -c . $O/counters_list.txt
grep -o "TCP_[A-Z_]*\|TA_[A-Z_]*\|TCC_[A-Z_]*\|SQ_[A-Z_]*" $O/counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
P3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
P4="TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
P5="SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
i=0
for cfg in "wino 1 400 64 64 32" "wino 1 50 512 512 128" "fp32 -1 50 512 512 128"; do
  i=$((i+1))
  for pass in 1 2 3 4 5; do
    eval C=\$P$pass
    (cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_${i}_${pass}" -o p -- python "$R/tools/one_layer.py" $cfg > "$R/$O/pmc_${i}_${pass}.log" 2>&1); echo "cfg $i ($cfg) pass $pass rc=$?"
  done
done
python - <<'PY'
import sqlite3, glob
for d in sorted(glob.glob("gpurun_out/r02c/pmc_*_*/")):
    for f in glob.glob(d + "**/*.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        try:
            rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%conv%kernel%' group by kernel_name, counter_name"))
        except Exception as e:
            print(d, "ERR", e); continue
        print(d)
        for r in rows: print("   %-34s %-34s %16.0f (n=%d)" % (r[0][:34], r[1], r[2], r[3]))
PY
