#!/bin/bash
# Round 2, call A: the whole GPU suite (with the new parity pins), then hardware MFMA-utilisation counters for the
# headline workload: two SQ passes over one step of bench.py (fp32 leg and split-precision leg).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
grep -h "structured\|b=.*vs b=2" $O/pytest_gpu.log | head -12
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for pass in 1 2; do
  if [ $pass = 1 ]; then C="$C1"; else C="$C2"; fi
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_sq_$pass" -o p -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$R/$O/pmc_sq_$pass.log" 2>&1); echo "pmc pass $pass rc=$?"
done
python tools/pmc_mfma.py $O/pmc_sq_1 $O/pmc_sq_2 > $O/pmc_mfma.json 2> $O/pmc_mfma.err; head -c 1500 $O/pmc_mfma.json; tail -3 $O/pmc_mfma.err
echo "== default bench"; timeout 600 python bench.py --cpu-seconds 6 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-400
