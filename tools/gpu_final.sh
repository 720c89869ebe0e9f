#!/bin/bash
# Round-end measurement: full GPU test suite, the default bench line, its rocprofv3 kernel-trace summary, the two PMC
# passes for HBM traffic, and the secondary configurations' bench lines.  Outputs under gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu (all)"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
echo "== default bench"; timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-250
echo "== rocprof default bench (kernel trace)"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_default" -o dflt -- python "$R/bench.py" --no-cpu-baseline > "$R/$O/rocprof_default.log" 2>&1); echo "rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"; (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_$C" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg > "$R/$O/pmc_$C.log" 2>&1); echo "rc=$?"
done
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | cut -c1-200; }
line train --mode train --steps 4 --warmup 1
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 5 --warmup 2
line resnet_h_train128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
line resnet_f_b32 --arch resnet_f --batch 32
line resnet_h_b128 --arch resnet_h --batch 128
line vgg_f_b32 --arch vgg_f --batch 32
line vgg_f_train32 --arch vgg_f --mode train --batch 32 --steps 3 --warmup 1
echo "== rocprof train"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_train" -o train -- python "$R/bench.py" --mode train --steps 2 --warmup 1 --no-cpu-baseline > "$R/$O/rocprof_train.log" 2>&1); echo "rc=$?"
echo "== rocprof resnet_h train16"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > "$R/$O/rocprof_rtrain.log" 2>&1); echo "rc=$?"
