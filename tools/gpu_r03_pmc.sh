#!/bin/bash
# after a bench.py edit: the two HBM-traffic PMC passes of the same file (roofline.traffic is keyed on its sha), then the default line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03final
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"; (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_$C" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg --no-secondary > "$R/$O/pmc_$C.log" 2>&1); echo "rc=$?"
done
python tools/pmc_traffic.py $(ls $O/pmc_FETCH_SIZE/*/*.db $O/pmc_FETCH_SIZE/*.db 2>/dev/null | head -1) $(ls $O/pmc_WRITE_SIZE/*/*.db $O/pmc_WRITE_SIZE/*.db 2>/dev/null | head -1) $O/pmc_traffic.json | tail -8
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
echo "== default bench"; timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
timeout 600 python bench.py --arch resnet_f --batch 32 --no-cpu-baseline --no-secondary > $O/bench_resnet_f_b32.log 2>&1; tail -1 $O/bench_resnet_f_b32.log | cut -c1-160
