#!/bin/bash
# Second box for the round-4 measurement set: the headline half (default bench under rocprofv3, the PMC traffic passes its
# `traffic` comes from, the default line) and the dispatch-timeline summaries restricted to the timed steps (marker kernel:
# tools/prof_summary.py <db> <out> <marker> <skip>).  Boxes differ by 4-6 % on unchanged kernels; profiles/README.md lists every run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04final_b
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
summ() { db=$(ls $1/*.db $1/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" "$2" "$3" "$4" > $O/summ.log 2>&1; echo "summary $2 rc=$?"; tail -7 $O/summ.log; rm -rf "$1"; }
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | cut -c1-160; }
echo "== microbench (unchanged kernels: the box's speed)"; timeout 300 python tools/microbench_wino4.py --batch 128 2>&1 | grep -v amdgpu.ids > $O/microbench_wino4_b128.txt; tail -1 $O/microbench_wino4_b128.txt
echo "== rocprof default bench (kernel trace)"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_default" -o dflt -- python "$R/bench.py" --no-cpu-baseline --no-secondary > "$R/$O/rocprof_default.log" 2>&1); echo "rc=$?"; grep -h '^{"metric' $O/rocprof_default.log | cut -c1-200
summ $O/prof_default $O/bench_default peaks_kernel 2
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"; (cd /tmp && DREAM_BENCH_PMC_CALIBRATE=1 timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_$C" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg --no-secondary > "$R/$O/pmc_$C.log" 2>&1); echo "rc=$?"
done
python tools/pmc_traffic.py $(ls $O/pmc_FETCH_SIZE/*/*.db $O/pmc_FETCH_SIZE/*.db 2>/dev/null | head -1) $(ls $O/pmc_WRITE_SIZE/*/*.db $O/pmc_WRITE_SIZE/*.db 2>/dev/null | head -1) $O/pmc_traffic.json | grep -i "ratio\|raw_to" | head
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json
echo "== default bench (with the PMC traffic of this bench.py)"; timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
echo "== rocprof train"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_train" -o train -- python "$R/bench.py" --mode train --steps 3 --warmup 2 --no-cpu-baseline > "$R/$O/rocprof_train.log" 2>&1); echo "rc=$?"
summ $O/prof_train $O/bench_train adam_kernel 1
echo "== rocprof resnet_h train16"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 5 --warmup 3 --no-cpu-baseline > "$R/$O/rocprof_rtrain.log" 2>&1); echo "rc=$?"
summ $O/prof_rtrain $O/bench_resnet_h_train16 adam_kernel 2
(cd /tmp && DREAM_BN_FUSION=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain3" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 5 --warmup 3 --no-cpu-baseline > "$R/$O/rocprof_rtrain3.log" 2>&1); echo "rc=$?"
summ $O/prof_rtrain3 $O/bench_resnet_h_train16_three_launch_bn adam_kernel 2
line train --mode train --steps 4 --warmup 1
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
line resnet_h_b128 --arch resnet_h --batch 128
du -sh $O
