#!/bin/bash
# Hardware MFMA utilisation (two SQ counter passes each) of the kernels that the default bench does not exercise: the ResNet
# inference path (transposed convs on the Winograd kernel, 1x1 GEMMs) and the two training paths (Winograd / GEMM weight
# gradients); plus the rocprofv3 kernel table of one GPU's share of configs[4].
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02pmc
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
run() { n=$1; shift
  for pass in 1 2; do
    if [ $pass = 1 ]; then C="$C1"; else C="$C2"; fi
    (cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/${n}_$pass" -o p -- python "$R/bench.py" "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg > "$R/$O/${n}_$pass.log" 2>&1); echo "$n pass $pass rc=$?"
  done
  python tools/pmc_mfma.py $O/${n}_1 $O/${n}_2 > $O/pmc_mfma_$n.json 2> $O/pmc_mfma_$n.err
  python -c "
import json; d=json.load(open('$O/pmc_mfma_$n.json'))['kernels']
for k,v in d.items(): print(' ', k, {a: round(b,3) for a,b in v.items() if a in ('dispatches','mfma_util','lds_conflict_frac','wait_frac','valu_insts_per_mfma')})"
  rm -rf $O/${n}_1 $O/${n}_2
}
run resnet_f_b32 --arch resnet_f --batch 32
run train --mode train
run resnet_h_train16 --arch resnet_h --mode train --batch 16
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rf" -o rf -- python "$R/bench.py" --arch resnet_f --batch 32 --no-cpu-baseline --no-split-leg > "$R/$O/rocprof_resnet_f.log" 2>&1); echo "rc=$?"
db=$(ls $O/prof_rf/*.db $O/prof_rf/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" $O/bench_resnet_f_b32 > /dev/null 2>&1; rm -rf $O/prof_rf
head -12 $O/bench_resnet_f_b32_kernel_stats.csv | cut -c1-150
du -sh $O
