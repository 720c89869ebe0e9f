#!/bin/bash
# Round-4 measurement set: full GPU suite, default bench line, rocprofv3 kernel-trace summary of the same command, the PMC
# passes for HBM traffic (FETCH_SIZE / WRITE_SIZE) and MFMA utilisation (SQ), ResNet training traffic, secondary lines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04final
mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu (all)"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
echo "== two ranks on this one GPU over gloo (rehearsal of the torchrun path; the numbers mean nothing)"
DREAM_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-split-leg > $O/rehearsal_2ranks_selflaunch.log 2>&1; echo "rc=$?"; tail -1 $O/rehearsal_2ranks_selflaunch.log | cut -c1-200
DREAM_BENCH_GPU_IDS=0,0,0,0 timeout 600 python bench.py --gpus 4 --single-process --arch resnet_h --mode train --steps 4 --warmup 3 --global-batch 32 --no-cpu-baseline > $O/rehearsal_single_process_train.log 2>&1; echo "rc=$?"; tail -1 $O/rehearsal_single_process_train.log | cut -c1-200
# raw rocprofv3 databases stay on the box (gpurun copies back at most 64 MiB): summarised here, then removed
summ() { db=$(ls $1/*.db $1/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" "$2" > /dev/null 2>&1; echo "summary $2 rc=$?"; rm -rf "$1"; }
echo "== rocprof default bench (kernel trace)"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_default" -o dflt -- python "$R/bench.py" --no-cpu-baseline --no-secondary > "$R/$O/rocprof_default.log" 2>&1); echo "rc=$?"; grep -h '^{"metric' $O/rocprof_default.log | cut -c1-200
summ $O/prof_default $O/bench_default
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"; (cd /tmp && DREAM_BENCH_PMC_CALIBRATE=1 timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_$C" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-split-leg --no-secondary > "$R/$O/pmc_$C.log" 2>&1); echo "rc=$?"
done
python tools/pmc_traffic.py $(ls $O/pmc_FETCH_SIZE/*/*.db $O/pmc_FETCH_SIZE/*.db 2>/dev/null | head -1) $(ls $O/pmc_WRITE_SIZE/*/*.db $O/pmc_WRITE_SIZE/*.db 2>/dev/null | head -1) $O/pmc_traffic.json | head -30
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for pass in 1 2; do
  if [ $pass = 1 ]; then C="$C1"; else C="$C2"; fi
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_sq_$pass" -o p -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > "$R/$O/pmc_sq_$pass.log" 2>&1); echo "pmc sq pass $pass rc=$?"
done
python tools/pmc_mfma.py $O/pmc_sq_1 $O/pmc_sq_2 > $O/pmc_mfma.json 2> $O/pmc_mfma.err; python -c "
import json; d=json.load(open('$O/pmc_mfma.json'))['kernels']
for k,v in d.items(): print(k, {a: round(b,3) for a,b in v.items() if a in ('mfma_util','lds_conflict_frac','wait_frac','issue_stall_frac','valu_insts_per_mfma')})"
rm -rf $O/pmc_sq_1 $O/pmc_sq_2
# ResNet training traffic (one GPU's share of configs[3])
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_rt_$C" -o pmc -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > "$R/$O/pmc_rt_$C.log" 2>&1); echo "resnet train pmc $C rc=$?"
done
python tools/pmc_traffic.py $(ls $O/pmc_rt_FETCH_SIZE/*/*.db $O/pmc_rt_FETCH_SIZE/*.db 2>/dev/null | head -1) $(ls $O/pmc_rt_WRITE_SIZE/*/*.db $O/pmc_rt_WRITE_SIZE/*.db 2>/dev/null | head -1) $O/pmc_traffic_resnet_train.json "bn_,gemm1x1_kernel,wgrad1x1,conv_mfma_kernel,conv_wino_kernel,conv_wino4_kernel,wgrad_kernel<,wgrad_wino,adam,pack" --arch resnet_h --mode train --batch 16 --steps 1 --warmup 1 | head -40
rm -rf $O/pmc_rt_FETCH_SIZE $O/pmc_rt_WRITE_SIZE
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | cut -c1-160; }
line train --mode train --steps 4 --warmup 1
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
DREAM_BN_FUSION=0 line resnet_h_train16_three_launch_bn --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
line resnet_h_train128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
DREAM_BN_FUSION=0 line resnet_h_train128_three_launch_bn --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
line resnet_f_b32 --arch resnet_f --batch 32
line resnet_h_b128 --arch resnet_h --batch 128
line vgg_f_b32 --arch vgg_f --batch 32
echo "== rocprof train"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_train" -o train -- python "$R/bench.py" --mode train --steps 2 --warmup 1 --no-cpu-baseline > "$R/$O/rocprof_train.log" 2>&1); echo "rc=$?"
summ $O/prof_train $O/bench_train
echo "== rocprof resnet_h train16"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > "$R/$O/rocprof_rtrain.log" 2>&1); echo "rc=$?"
summ $O/prof_rtrain $O/bench_resnet_h_train16
echo "== layer profiles"
for cfg in "resnet_h train 16" "vgg_q infer 128"; do set -- $cfg
  timeout 300 python tools/layer_profile.py --arch $1 --mode $2 --batch $3 --top 45 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_$1_$2$3.txt; head -3 $O/layer_profile_$1_$2$3.txt | cut -c1-200
done
echo "== microbenches"; timeout 300 python tools/microbench_wino4.py --batch 128 2>&1 | grep -v amdgpu.ids > $O/microbench_wino4_b128.txt; tail -1 $O/microbench_wino4_b128.txt
timeout 300 python tools/ab_wino4_pinning.py 2>&1 | grep -v amdgpu.ids > $O/ab_wino4_pinning.txt; tail -1 $O/ab_wino4_pinning.txt
timeout 300 python tools/microbench_conv1x1.py --batch 16 2>&1 | grep -v amdgpu.ids > $O/microbench_conv1x1_b16.txt; tail -2 $O/microbench_conv1x1_b16.txt
echo "== L2 hit rate of the F(4x4) kernel, channel blocks walked / pinned (512->512 @ 50x50, b=128)"
for pin in 0 1; do
  (cd /tmp && DREAM_W4_YMAP=$pin timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace -d "$R/$O/l2_$pin" -o p -- python "$R/tools/one_layer.py" wino4 0 50 512 512 128 3 > "$R/$O/l2_$pin.log" 2>&1); echo "l2 pass pin=$pin rc=$?"
done
python - <<'PY' | tee $O/l2_hit_wino4_pinning.txt
import sqlite3, glob
for pin in (0, 1):
    for f in glob.glob("gpurun_out/r04final/l2_%d/**/*.db" % pin, recursive=True):
        cur = sqlite3.connect(f).cursor()
        rows = dict((r[0], r[1]) for r in cur.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%conv_wino4_kernel%' group by counter_name"))
        hit, miss = rows.get("TCC_HIT_sum", 0), rows.get("TCC_MISS_sum", 0)
        print("channel blocks %s: per launch TCC_HIT %.3e TCC_MISS %.3e hit rate %.3f, TCP->TCC read requests %.3e" % (
            "pinned to XCDs" if pin else "walked by every XCD", hit, miss, hit / max(hit + miss, 1), rows.get("TCP_TCC_READ_REQ_sum", 0)))
PY
rm -rf $O/l2_0 $O/l2_1
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json
echo "== default bench again (now with the PMC traffic of this bench.py)"; timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
ls -la $O | head -60; du -sh $O
