#!/bin/bash
# ONE parameterised GPU-box script (replaces the per-round gpu_rNN_* families): tools/gpu_round.sh <stage> [args].
# Run through gpurun from the repo root; writes under gpurun_out/r06_<stage>/ (round 5's stages wrote r05_<stage>).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_$1
mkdir -p $O
export TMPDIR=/tmp
PROD=dream_amd/libdream_hip.so
line() { n=$1; shift; timeout ${LINE_TIMEOUT:-600} python bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
    print('$n', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms', 'frac', round(r.get('frac') or 0,4), 'exec', r.get('executed_frac'))
except Exception as e: print('$n', 'FAILED', e)
"; }
withlib() { cp $PROD /tmp/lib_keep.so; cp build/lib$1.so $PROD; shift; "$@"; cp /tmp/lib_keep.so $PROD; }
case "$1" in
tests)
  # the whole GPU suite + smoke() + the timing probe of the data-parallel steps
  echo "== pytest gpu (all)"; timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-60,100-190
  timeout 1200 python tools/dp_exchange_probe.py 3 > $O/dp_exchange_probe.txt 2> $O/dp_exchange_probe.err; echo "probe rc=$?"; tail -1 $O/dp_exchange_probe.txt
  ;;
g6i)
  # Round 6: the one-launch weight re-pack with its workgroups dealt out by job size (DREAM_PACK_SPANS=0: 16 per job as before)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "clone or pack or variant or ms2 or skip or golden or train" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    DREAM_PACK_SPANS=0 line spans0_$r $R
    line spans1_$r $R
  done
  DREAM_PACK_SPANS=0 line rf32_spans0 --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  line rf32_spans1 --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_pack -o t -- python $GRAFT_REPO_ROOT/bench.py $R --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1)
  python - $(ls /tmp/prof_pack/*.db /tmp/prof_pack/*/*.db 2>/dev/null | head -1) <<'PY' | tee $O/pack_kernels.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for name, n, tot, mn in cur.execute("select name, count(*), sum(duration), min(duration) from kernels where name like '%pack%' group by name"):
    print("%-90s launches %4d  avg %8.1f us  min %8.1f us" % (name.split("(")[0][-90:], n, tot / n / 1e3, mn / 1e3))
PY
  ;;
g6j)
  # Round 6: the last decoder BatchNorm's masked backward sums in the head conv's data gradient (DREAM_BN_FUSION_HEAD=0: stand-alone pass)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "resnet or conv1x1 or bn or data_parallel or probe or synchronisation" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    DREAM_BN_FUSION_HEAD=0 line head0_$r $R
    line head1_$r $R
  done
  DREAM_BN_FUSION_HEAD=0 line rt128_head0 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  line rt128_head1 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  for r in a b; do
    DREAM_BN_FUSION_HEAD=0 line rf32_head0_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
    line rf32_head1_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  done
  ;;
g6y)
  # Round 6: evaluation stem (im2col rows, K = 160) on the 1x1 GEMM (DREAM_STEM_GEMM=0 = the 1-tap direct kernel); the transposed-conv GEMM without a K split
  echo "== pytest"; timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest.log | head
  for r in a b c; do
    DREAM_STEM_GEMM=0 line rf32_old_$r --arch resnet_f --batch 32
    line rf32_new_$r --arch resnet_f --batch 32
  done
  for r in a b; do
    DREAM_STEM_GEMM=0 line rh128_old_$r --arch resnet_h --batch 128
    line rh128_new_$r --arch resnet_h --batch 128
  done
  line rt16 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  ;;
g6x)
  # Round 6: the evaluation path: stride-2 downsample convs on the GEMM over gathered pixels, the first decoder layer's transposed conv as GEMM + gather
  # (folded BatchNorm + ReLU in the gather) -- DREAM_DS_GEMM=0 = the direct / Winograd kernels
  echo "== pytest"; timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest.log | head
  for r in a b c; do
    DREAM_DS_GEMM=0 line rf32_old_$r --arch resnet_f --batch 32
    line rf32_new_$r --arch resnet_f --batch 32
  done
  for r in a b; do
    DREAM_DS_GEMM=0 line rh128_old_$r --arch resnet_h --batch 128
    line rh128_new_$r --arch resnet_h --batch 128
  done
  ;;
g6w)
  # Round 6: the decoder's transposed convs on small maps as ONE 1x1 GEMM (N = 16 Cout) + a gather (training forward): DREAM_CONVT_GEMM_MAX_PIXELS=0
  # (Winograd kernel) / 4096 (default: the first decoder layer at 16 frames) / 12000 (+ the second)
  echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -k "resnet or layouts or pool or train" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    DREAM_CONVT_GEMM_MAX_PIXELS=0 line rt16_t0_$r $R
    line rt16_t4096_$r $R
    DREAM_CONVT_GEMM_MAX_PIXELS=12000 line rt16_t12000_$r $R
  done
  for r in a b; do
    DREAM_CONVT_GEMM_MAX_PIXELS=0 line rf32_t0_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
    DREAM_CONVT_GEMM_MAX_PIXELS=6000 line rf32_t6000_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  done
  ;;
g6v)
  # Round 6: 3x3 stride-2 convs with few output pixels on the 1x1 GEMM over their patch rows (im2col3s2 / col2im3s2): DREAM_COL3_MAX_PIXELS=0 (direct
  # kernels) / 4096 (default: layer4.0.conv2 at 16 frames) / 12000 (+ layer3.0.conv2)
  echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -k "resnet or layouts or pool or train" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    DREAM_COL3_MAX_PIXELS=0 line rt16_col0_$r $R
    line rt16_col4096_$r $R
    DREAM_COL3_MAX_PIXELS=12000 line rt16_col12000_$r $R
  done
  for r in a b; do
    DREAM_COL3_MAX_PIXELS=0 line rf32_col0_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
    DREAM_COL3_MAX_PIXELS=6000 line rf32_col6000_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  done
  ;;
g6u)
  # Round 6: the Winograd-domain weight gradient on 64 x 64-channel layers from 150 k pixels on (was: 4 M) -- DREAM_WGRAD_WINOGRAD_MIN_PIXELS=4000000 = before
  echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -k "wgrad or train or full_size or headline" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for r in a b c; do
    DREAM_WGRAD_WINOGRAD_MIN_PIXELS=4000000 line vq_train_old_$r --mode train --steps 5 --warmup 2
    line vq_train_new_$r --mode train --steps 5 --warmup 2
  done
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    DREAM_WGRAD_WINOGRAD_MIN_PIXELS=4000000 line rt16_old_$r $R
    line rt16_new_$r $R
  done
  ;;
g6t)
  # Round 6: the trunk's three stride-2 downsample convs on the 1x1 GEMM over gathered pixels (forward with the BatchNorm statistics in the epilogue,
  # weight gradient, data gradient + scatter) against the direct kernels + statistics pass (DREAM_DS_GEMM=0)
  echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -k "resnet or layouts or pool or data_parallel or graph or train" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    DREAM_DS_GEMM=0 line rt16_ds0_$r $R
    line rt16_ds1_$r $R
  done
  DREAM_DS_GEMM=0 line rt128_ds0 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  line rt128_ds1 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  for r in a b; do
    DREAM_DS_GEMM=0 line rf32_ds0_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
    line rf32_ds1_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  done
  timeout 300 python tools/layer_profile.py --arch resnet_h --mode train --batch 16 --top 60 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_resnet_h_train16.txt; grep -n "subsample2\|scatter2\|conv2d" $O/layer_profile_resnet_h_train16.txt | head -20 | cut -c1-160
  ;;
g6s)
  # Round 6: Winograd-domain weight gradients, step 4: ONE dy load per stage (a lane owns one pixel of the tile's 2 x 2; the other column / row by DPP;
  # row 2 of A dY A^T stored negated, flipped by the reduction kernels) -- build/libwgw_new4.so against build/libhead.so (= the commit before)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "convT or wgrad or train" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for v in head wgw_new4; do
    withlib $v timeout 300 python tools/microbench_convT_wgrad.py --digest --layers 5 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/digest_convT_$v.txt
    withlib $v timeout 300 python tools/microbench_wgrad_wino.py --digest --batch 16 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/digest_3x3_$v.txt
  done
  echo "digest lines that differ (dw must not; db may: its order of additions changed):"; diff $O/digest_convT_head.txt $O/digest_convT_wgw_new4.txt | head -12; diff $O/digest_3x3_head.txt $O/digest_3x3_wgw_new4.txt | head -26
  for r in a b; do for v in head wgw_new4; do
    echo "-- $v $r"; withlib $v timeout 300 python tools/microbench_convT_wgrad.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/microbench_convT_$v.txt | cut -c88-160
  done; done
  for r in a b; do for v in head wgw_new4; do
    echo "-- 3x3 $v $r"; withlib $v timeout 400 python tools/microbench_wgrad_wino.py --only-winograd 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/microbench_3x3_$v.txt | tail -1
  done; done
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    withlib head line rt16_old_$r $R
    withlib wgw_new4 line rt16_new_$r $R
  done
  for r in a b; do
    withlib head line vq_train_old_$r --mode train --steps 5 --warmup 2
    withlib wgw_new4 line vq_train_new_$r --mode train --steps 5 --warmup 2
  done
  ;;
g6r)
  # Round 6: 1x1 GEMM epilogues of the statistics forms: no scale / ReLU arithmetic where the entry points pass none, no row mask in the masked
  # data gradient (rows past M are loaded as zeros), its two sums in fp32 over the lane's rows (fp64 over lanes, waves and tiles) --
  # build/libg1_new.so against build/libcw_new.so (= the commit before)
  echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -k "conv1x1 or resnet or bn or train or data_parallel or probe" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for v in cw_new g1_new; do
    echo "-- $v"; withlib $v timeout 300 python tools/microbench_conv1x1.py --batch 16 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/microbench_conv1x1_$v.txt | tail -12 | cut -c1-200
  done
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    withlib cw_new line rt16_old_$r $R
    withlib g1_new line rt16_new_$r $R
  done
  withlib cw_new line rt128_old --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  withlib g1_new line rt128_new --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  ;;
g6q)
  # Round 6: F(2x2,3x3) forward / data-gradient kernels (conv_wino_body.inc): the column transform's quad exchange folded into v_fmac_f32 with a DPP
  # source (16 of the loop's 50 VALU instructions per 72 MFMAs) -- build/libcw_new.so against build/libwgw_new3.so (= the commit before)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "winograd or resnet or structured or vgg_f" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for v in wgw_new3 cw_new; do
    for cfg in "resnet_h infer 16" "resnet_h train 16" "vgg_f infer 8"; do set -- $cfg
      withlib $v timeout 300 python tools/digest_step.py --arch $1 --mode $2 --batch $3 2>&1 | grep sha256 >> $O/digest_$v.txt
    done
  done
  cmp $O/digest_wgw_new3.txt $O/digest_cw_new.txt && echo "digests identical"; cat $O/digest_cw_new.txt
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    withlib wgw_new3 line rt16_old_$r $R
    withlib cw_new line rt16_new_$r $R
  done
  for r in a b; do
    withlib wgw_new3 line ri16_old_$r --arch resnet_h --batch 16
    withlib cw_new line ri16_new_$r --arch resnet_h --batch 16
    withlib wgw_new3 line vf32_old_$r --arch vgg_f --batch 32
    withlib cw_new line vf32_new_$r --arch vgg_f --batch 32
  done
  ;;
g6p)
  # Round 6: weight-gradient kernels in the Winograd domain (3x3 and CONVT forms), step 3: one v_add per load (separate out-of-range words for the
  # dead row / dead column), the V row transform as v_fmac_f32 with a DPP source, the dM row transform as one fma with a lane constant --
  # build/libwgw_new3.so against step 2 (build/libwgw_new2.so) and the tree before (build/libwgw_old.so)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "convT or wgrad or train" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for v in old new3; do
    withlib wgw_$v timeout 300 python tools/microbench_convT_wgrad.py --digest --layers 5 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/digest_convT_$v.txt
    withlib wgw_$v timeout 300 python tools/microbench_wgrad_wino.py --digest --batch 16 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/digest_3x3_$v.txt
  done
  cmp $O/digest_convT_old.txt $O/digest_convT_new3.txt && echo "convT digests identical"
  cmp $O/digest_3x3_old.txt $O/digest_3x3_new3.txt && echo "3x3 digests identical"
  for r in a b; do for v in new2 new3; do
    echo "-- $v $r"; withlib wgw_$v timeout 300 python tools/microbench_convT_wgrad.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/microbench_convT_$v.txt | cut -c88-160
  done; done
  for r in a b; do for v in old new3; do
    echo "-- 3x3 $v $r"; withlib wgw_$v timeout 400 python tools/microbench_wgrad_wino.py --only-winograd 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/microbench_3x3_$v.txt
  done; done
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    withlib wgw_old line rt16_old_$r $R
    withlib wgw_new3 line rt16_new3_$r $R
  done
  for r in a b; do
    withlib wgw_old line vq_train_old_$r --mode train --steps 5 --warmup 2
    withlib wgw_new3 line vq_train_new3_$r --mode train --steps 5 --warmup 2
  done
  ;;
g6o)
  # Round 6: CONVT weight gradient, step 2: the shared position's operands read as the components the wave uses (no selects) -- build/libwgw_new2.so
  # against step 1 (build/libwgw_new.so) and the tree before (build/libwgw_old.so)
  for v in old new2; do withlib wgw_$v timeout 300 python tools/microbench_convT_wgrad.py --digest --layers 5 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/digest_$v.txt; done
  cmp $O/digest_old.txt $O/digest_new2.txt && echo "digests identical"
  for r in a b; do for v in new new2; do
    echo "-- $v $r"; withlib wgw_$v timeout 300 python tools/microbench_convT_wgrad.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/microbench_$v.txt | cut -c88-200
  done; done
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    withlib wgw_old line rt16_old_$r $R
    withlib wgw_new2 line rt16_new2_$r $R
  done
  ;;
g6n)
  # Round 6: ConvTranspose2d weight gradient (nine positions) with the phase column compile-time: five loads per stage instead of six, no operand
  # selects, the dead patch row's lanes masked (build/libwgw_old.so = before).  Digests: bit-identity of the two builds.
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "convT or wgrad or resnet or train" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for v in old new; do withlib wgw_$v timeout 300 python tools/microbench_convT_wgrad.py --digest --layers 5 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/digest_$v.txt; done
  cmp $O/digest_old.txt $O/digest_new.txt && echo "digests identical"; cat $O/digest_new.txt
  for r in a b; do for v in old new; do
    echo "-- $v $r"; withlib wgw_$v timeout 300 python tools/microbench_convT_wgrad.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/microbench_$v.txt | cut -c1-200
  done; done
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    withlib wgw_old line rt16_old_$r $R
    withlib wgw_new line rt16_new_$r $R
  done
  withlib wgw_old line rt128_old --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  withlib wgw_new line rt128_new --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  withlib wgw_old line rf32_old --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  withlib wgw_new line rf32_new --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  ;;
pk2)
  # Round 6: peak-extraction kernels with the loads of a strip in flight together and 32-bit offsets (build/libpeaks_old.so = before)
  echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 -k "peaks or abi or gauss or structured" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for r in a b; do
    for v in old new; do
      echo "-- $v $r"; withlib peaks_$v timeout 200 python tools/microbench_peaks.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/microbench_peaks_$v.txt
    done
  done
  for r in a b; do
    withlib peaks_old line rf32_old_$r --arch resnet_f --batch 32
    withlib peaks_new line rf32_new_$r --arch resnet_f --batch 32
  done
  ;;
pk)
  # Round 6: hardware counters of the peak-extraction kernels inside the resnet_f inference step (544 maps of 416 x 416)
  R="$PWD"
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE"
  P3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
  for pass in 1 2 3; do
    eval C=\$P$pass
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_$pass" -o p -- python "$R/bench.py" --arch resnet_f --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > "$R/$O/pmc_$pass.log" 2>&1); echo "pmc pass $pass rc=$?"
  done
  python tools/pmc_kernels.py gauss,peaks $O/pmc_* > $O/pmc_peaks.txt 2> $O/pmc.err; cat $O/pmc_peaks.txt | cut -c1-400
  rm -rf $O/pmc_[1-9]
  ;;
g6m)
  # Round 6: the step's weight re-pack in two launches, the late one (layer3 on, decoder, data-gradient operators) on the second stream beside the
  # forward pass of the stem, layer1 and layer2 (DREAM_PACK_SPLIT=1; measured: no gain, opt-in) against one launch on the main stream
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "packing or resnet or data_parallel or train or graph" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    line split0_$r $R
    DREAM_PACK_SPLIT=1 line split1_$r $R
  done
  for r in a b; do
    line rf32_split0_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
    DREAM_PACK_SPLIT=1 line rf32_split1_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  done
  line rt128_split0 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  DREAM_PACK_SPLIT=1 line rt128_split1 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  ;;
g6l)
  # Round 6: the 7x7 stem on the 1x1 GEMM (im2col rows of 192 columns, BatchNorm statistics in its epilogue, GEMM-shaped weight gradient)
  # against the direct kernels on 160 columns (DREAM_STEM_GEMM=0)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "resnet or conv1x1 or bn or data_parallel or train" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  for r in a b c; do
    DREAM_STEM_GEMM=0 line stem0_$r $R
    line stem1_$r $R
  done
  DREAM_STEM_GEMM=0 line rt128_stem0 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  line rt128_stem1 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  for r in a b; do
    DREAM_STEM_GEMM=0 line rf32_stem0_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
    line rf32_stem1_$r --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  done
  ;;
g6k)
  # Round 6: belief-map peak extraction with the second Gaussian pass fused with the scan (DREAM_PEAKS_FUSED=0: three kernels)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "peaks or abi or smoke or structured or inference or golden" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for r in a b c; do
    DREAM_PEAKS_FUSED=0 line rf32_three_$r --arch resnet_f --batch 32
    line rf32_fused_$r --arch resnet_f --batch 32
  done
  for r in a b; do
    DREAM_PEAKS_FUSED=0 line vq128_three_$r
    line vq128_fused_$r
  done
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_pk -o t -- python $GRAFT_REPO_ROOT/bench.py --arch resnet_f --batch 32 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1)
  python - $(ls /tmp/prof_pk/*.db /tmp/prof_pk/*/*.db 2>/dev/null | head -1) <<'PY' | tee $O/peaks_kernels.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for name, n, tot, mn in cur.execute("select name, count(*), sum(duration), min(duration) from kernels where name like '%gauss%' or name like '%peaks%' group by name"):
    print("%-60s launches %4d  avg %8.1f us  min %8.1f us" % (name.split("(")[0][-60:], n, tot / n / 1e3, mn / 1e3))
PY
  ;;
lp)
  # layer profiles with queued event pairs (no launch latency inside the measurements) against the per-call synchronisation of rounds 2-6
  for t in queued sync; do
    timeout 300 python tools/layer_profile.py --arch resnet_h --mode train --batch 16 --top 45 --timing $t 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_resnet_h_train16_$t.txt; head -16 $O/layer_profile_resnet_h_train16_$t.txt | cut -c1-120
  done
  for cfg in "vgg_q infer 128" "vgg_q train 128" "resnet_f infer 32"; do set -- $cfg
    timeout 300 python tools/layer_profile.py --arch $1 --mode $2 --batch $3 --top 45 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_$1_$2$3.txt; head -3 $O/layer_profile_$1_$2$3.txt | cut -c1-120
  done
  ;;
nodes)
  # do the captured steps still hold memset / memcpy nodes?  (the runtime executes them as __amd_rocclr_* blit kernels: count those in the
  # kernel traces of 4 and of 10 replayed steps -- the difference / 6 is what ONE replayed step holds; such nodes are not reliably ordered
  # with the kernel nodes around them on this runtime -- tools/dp_exchange_probe.py)
  cd /tmp
  for w in ${2:+"resnet_h train 16" "vgg_q train 32"} "vgg_q inference 32 --graph" "resnet_h inference 16 --graph"; do
    set -- $w
    for k in 4 10; do
      rm -rf /tmp/prof_nodes
      DREAM_TRAIN_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_nodes -o t -- python $GRAFT_REPO_ROOT/bench.py --arch $1 --mode $2 --batch $3 $4 --steps $k --warmup 3 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/run_$1_$2_$k.log 2>&1
      db=$(ls /tmp/prof_nodes/*.db /tmp/prof_nodes/*/*.db 2>/dev/null | head -1)
      python - "$db" "$w" $k <<'PY' | tee -a $GRAFT_REPO_ROOT/$O/graph_nodes.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, count(*) from kernels group by name"))
tot = sum(n for _, n in rows)
blit = sorted((k.split("(")[0], n) for k, n in rows if "rocclr" in k)
print("%-28s %2s timed steps: kernels traced %6d; runtime blit kernels: %s" % (sys.argv[2], sys.argv[3], tot, blit or "none"))
PY
    done
  done
  ;;
probe)
  # Round 6: do the replayed data-parallel training steps depend on when the GPU runs them?  (tools/dp_exchange_probe.py; $2 = runs a setting,
  # $3 = environment settings of the HIP runtime to try, e.g. "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0")
  for v in ${3:-""}; do
    echo "== env: $v"
    env $v timeout 1200 python tools/dp_exchange_probe.py ${2:-3} > $O/dp_exchange_probe_$v.txt 2> $O/dp_exchange_probe.err; echo "rc=$?"; cut -c1-75,215- $O/dp_exchange_probe_$v.txt
  done
  ;;
g6g)
  # Round 6, verdict task 2b: the single-process exchange in two pieces (early bucket behind each replica's event on an exchange stream)
  echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -k "data_parallel or allreduce or bucketed or one_device_training or graph" -s > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  for r in a b; do
    DREAM_BENCH_GPU_IDS=0,0,0,0 timeout 600 python bench.py --gpus 4 --single-process --arch resnet_h --mode train --steps 6 --warmup 3 --global-batch 64 --no-cpu-baseline > $O/dp4_buckets_$r.log 2>&1; tail -1 $O/dp4_buckets_$r.log | cut -c1-160
    DREAM_DP_BUCKETS=0 DREAM_BENCH_GPU_IDS=0,0,0,0 timeout 600 python bench.py --gpus 4 --single-process --arch resnet_h --mode train --steps 6 --warmup 3 --global-batch 64 --no-cpu-baseline > $O/dp4_one_piece_$r.log 2>&1; tail -1 $O/dp4_one_piece_$r.log | cut -c1-160
  done
  DREAM_FORCE_RCCL=1 DREAM_TRAIN_GRAPH=1 line rt16_graph_rccl_buckets --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  DREAM_TRAIN_GRAPH=1 line rt16_graph --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  line rt16_eager --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  ;;
gcx)
  # the occasional long step of the ResNet-101 training run (one step of ten at +16-19 ms in 2 of the final set's 10 runs): is it the cyclic GC?
  # interleaved runs: default / DREAM_BENCH_GC=off / DREAM_BENCH_GC=freeze
  for i in $(seq 1 ${2:-14}); do for m in default off freeze; do
    if [ $m = default ]; then E=""; else E="DREAM_BENCH_GC=$m"; fi
    env $E timeout 300 python bench.py --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/run_${m}_$i.log 2>&1
    tail -1 $O/run_${m}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['step_ms']
print('$m run $i %.1f frames/s  gpu step ms min %.2f median %.2f max %.2f  host median %.2f max %.2f' % (d['value'], s['gpu_min'], s['gpu_median'], s['gpu_max'], s['host_median'], s['host_max']))" | tee -a $O/runs.txt
  done; done
  ;;
slow)
  # the occasional slow run of the ResNet-101 training step at 16 frames: N consecutive runs, per-step GPU / host times from bench.py's
  # end-of-step events (one long step, or a uniformly slower run?); $2 = extra environment (e.g. PYTHONGC=off handled by bench.py)
  for i in $(seq 1 ${3:-16}); do
    env $2 timeout 300 python bench.py --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/run_$i.log 2>&1
    tail -1 $O/run_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['step_ms']
print('run $i %.1f frames/s  gpu step ms min %.2f median %.2f max %.2f  host median %.2f max %.2f' % (d['value'], s['gpu_min'], s['gpu_median'], s['gpu_max'], s['host_median'], s['host_max']))"
  done
  ;;
g6f)
  # Round 6: 32-row wavefront tiles of the 1x1 GEMM where 64-row tiles leave the SIMDs with fewer than four each (default) against 64-row tiles
  # everywhere (DREAM_CONV1X1_ROWS=64)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "conv1x1 or resnet or bn_fused or stress" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  timeout 400 python tools/microbench_gemm_forms.py --shapes layer3,layer2,layer4 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/gemm_forms_rows_auto.txt
  DREAM_CONV1X1_ROWS=64 timeout 400 python tools/microbench_gemm_forms.py --shapes layer3,layer2,layer4 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/gemm_forms_rows64.txt
  for r in a b c; do
    line rt16_auto_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_CONV1X1_ROWS=64 line rt16_rows64_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  line rh16_auto --arch resnet_h --batch 16
  DREAM_CONV1X1_ROWS=64 line rh16_rows64 --arch resnet_h --batch 16
  line rf32_auto --arch resnet_f --batch 32
  DREAM_CONV1X1_ROWS=64 line rf32_rows64 --arch resnet_f --batch 32
  DREAM_CONV1X1_ROWS=32 line rf32_rows32 --arch resnet_f --batch 32
  line rh128_auto --arch resnet_h --batch 128
  DREAM_CONV1X1_ROWS=32 line rh128_rows32 --arch resnet_h --batch 128
  ;;
g6e)
  # Round 6: the weight gradient of an upsample + conv3x3 on the nine-position transposed-conv form (default) against the sixteen-position
  # kernel with the fused upsample (DREAM_UPS_WGRAD=winograd16); vgg_q / vgg_f training; the new structured fixture; advice fixes
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "wgrad_winograd or train_step or train_steps or reference_golden or structured or hip_graph or one_device_training or variant" -s > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  for r in a b c; do
    line vt_convT9_$r --mode train --steps 5 --warmup 2
    DREAM_UPS_WGRAD=winograd16 line vt_wino16_$r --mode train --steps 5 --warmup 2
  done
  ;;
g6d)
  # Round 6: width of the weight-gradient launches on the second stream (DREAM_SIDE_WGRAD_WIDTH, per cent of the full split-K workgroup count):
  # narrow launches leave compute units to the data-gradient chain instead of taking the whole chip in bursts
  for r in a b; do for w in 100 60 40 25 15; do
    DREAM_SIDE_WGRAD_WIDTH=$w line rt16_w${w}_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done; done
  for w in 100 40 25; do DREAM_SIDE_WGRAD_WIDTH=$w line rft32_w${w} --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2; done
  for w in 100 40 25; do DREAM_SIDE_WGRAD_WIDTH=$w line rt64_w${w} --arch resnet_h --mode train --batch 64 --steps 4 --warmup 2; done
  ;;
prof16)
  # where the ResNet-101 training step at 16 frames goes now: rocprofv3 kernel table + dispatch timeline, the synchronised layer profile
  R="$PWD"
  summ() { db=$(ls $1/*.db $1/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" "$2" $3 $4 > $O/summ.log 2>&1; echo "summary $2 rc=$?"; rm -rf "$1"; }
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 8 --warmup 3 --no-cpu-baseline > "$R/$O/rocprof_rtrain.log" 2>&1); echo "rc=$?"
  summ $O/prof_rtrain $O/bench_resnet_h_train16 adam_kernel 3
  timeout 400 python tools/layer_profile.py --arch resnet_h --mode train --batch 16 --top 60 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_resnet_h_train16.txt; head -40 $O/layer_profile_resnet_h_train16.txt | cut -c1-60,100-190
  cat $O/bench_resnet_h_train16_concurrency.txt
  ;;
rehearse8)
  # Round 6, verdict task 7: both multi-GPU paths at the REAL rank count (8) on the one GPU of the box -- the numbers mean nothing, the
  # code path (self-launch under torch.distributed.run, 8 ranks, the `scale` block with configs[3] 8 x 16 frames and configs[4] 8 x 32 frames
  # at K = 17; 8 replicas in one process, split graphs, one exchange) is what the first 8-GPU node will run.
  DREAM_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 2 --warmup 1 --secondary-steps 2 --secondary-train-steps 3 > $O/rehearsal_8ranks_gloo.log 2>&1; echo "rc=$?"; tail -1 $O/rehearsal_8ranks_gloo.log | cut -c1-600
  DREAM_BENCH_GPU_IDS=0,0,0,0,0,0,0,0 timeout 900 python bench.py --gpus 8 --single-process --arch resnet_h --mode train --steps 4 --warmup 3 --global-batch 128 --no-cpu-baseline > $O/rehearsal_single_process_8replicas_train.log 2>&1; echo "rc=$?"; tail -1 $O/rehearsal_single_process_8replicas_train.log | cut -c1-600
  DREAM_BENCH_GPU_IDS=0,0,0,0,0,0,0,0 timeout 900 python bench.py --gpus 8 --single-process --arch resnet_f --steps 3 --warmup 2 --global-batch 256 --no-cpu-baseline > $O/rehearsal_single_process_8replicas_infer.log 2>&1; echo "rc=$?"; tail -1 $O/rehearsal_single_process_8replicas_infer.log | cut -c1-600
  ;;
g6c)
  # Round 6, verdict task 4c: conv1_1 -> conv1_2 (+ pool) over sub-batches (the 64 x 400 x 400 tensor between them stays in the Infinity Cache)
  echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "sub_batches or headline_batch" -s > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  for r in a b c; do
    line dflt_whole_$r
    for n in 4 6 8 16; do DREAM_FIRST_SUBBATCH=$n line dflt_sub${n}_$r; done
  done
  ;;
g6b)
  # Round 6, verdict task 1b: ConvTranspose2d(4,2,1) weight gradient on the nine-position F(2x2,2x2) form against the direct kernel
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "convT4x4_wgrad or resnet or train_step or backward_ops or data_parallel" -s > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log; grep "convT4x4 wgrad" $O/pytest.log
  timeout 300 python tools/microbench_convT_wgrad.py --batch 16 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/microbench_convT_wgrad_b16.txt
  timeout 300 python tools/microbench_convT_wgrad.py --batch 128 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/microbench_convT_wgrad_b128.txt
  timeout 300 python tools/microbench_convT_wgrad.py --batch 32 --layers 5 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/microbench_convT_wgrad_b32.txt
  for r in a b c; do
    line rt16_wino_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_CONVT_WGRAD=direct line rt16_direct_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  for r in a b; do
    line rt128_wino_$r --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
    DREAM_CONVT_WGRAD=direct line rt128_direct_$r --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  done
  line rft32_wino --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  DREAM_CONVT_WGRAD=direct line rft32_direct --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  ;;
g6a)
  # Round 6, verdict task 1a: the 1x1 GEMMs of ResNet-101 IN the training step.  (1) parity of the re-scheduled kernels (pinned loads,
  # deep epilogue prefetch), (2) every form of the GEMM on warm and on cold operands, new library and round 5's (build/libR5.so),
  # (3) hardware counters of the gemm1x1 / wgrad1x1 launches inside one training step (PMC passes serialise the dispatches: the
  # kernel's own behaviour on the step's cold operands, without the second stream), (4) training / inference lines, alternating.
  R="$PWD"
  echo "== pytest (1x1 GEMM forms, BatchNorm folding, ResNet)"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "conv1x1 or resnet or bn_fused or stress or headline_batch or full_size_batch" -s > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  echo "== forms, new library"; timeout 400 python tools/microbench_gemm_forms.py --shapes layer3,layer2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/gemm_forms_new.txt
  echo "== forms, round-5 library"; withlib R5 timeout 400 python tools/microbench_gemm_forms.py --shapes layer3,layer2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/gemm_forms_r5.txt
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  P2="SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SALU"
  P3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
  P4="TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_EA0_RDREQ_sum"
  pmc() { # pmc <tag> <passes...>
    tag=$1; shift
    for pass in "$@"; do
      eval C=\$P$pass
      (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_${tag}_$pass" -o p -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > "$R/$O/pmc_${tag}_$pass.log" 2>&1); echo "pmc $tag pass $pass rc=$?"
    done
    python tools/pmc_kernels.py gemm1x1_kernel,wgrad1x1_kernel $O/pmc_${tag}_* > $O/pmc_gemm1x1_in_step_$tag.txt 2> $O/pmc_$tag.err; head -30 $O/pmc_gemm1x1_in_step_$tag.txt | cut -c1-400
    rm -rf $O/pmc_${tag}_[1-9]
  }
  echo "== PMC in the step, round-5 library"; withlib R5 pmc r5 1 2 3 4
  echo "== PMC in the step, new library"; pmc new 1 2 3 4
  for r in a b c; do
    line rt16_new_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    withlib R5 line rt16_R5_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  line rh128_new --arch resnet_h --batch 128
  withlib R5 line rh128_R5 --arch resnet_h --batch 128
  line rf32_new --arch resnet_f --batch 32
  withlib R5 line rf32_R5 --arch resnet_f --batch 32
  line rt128_new --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  withlib R5 line rt128_R5 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  ;;
ab1)
  # packed vs scalar fp32 VALU beside the MFMAs (verdict round 4, task 1a), running weight offset, s_setprio
  echo "== pytest on the scalar build"; withlib SC timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "winograd4 or wino4 or structured or pinned or conv_winograd" > $O/pytest_sc.log 2>&1; echo "rc=$?"; tail -2 $O/pytest_sc.log
  echo "== wino4 per-layer round-robin"; DREAM_W4_DIAG_KS=4001,4002,4003,4004,4005,15 timeout 400 python tools/wino4_diag.py run --batch 128 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/wino4_diag.txt
  echo "== headline, library A/B"
  for l in R4 PK SC SCP PK SC R4 SCP; do withlib $l line dflt_$l; done
  echo "== training, library A/B"
  for l in PK SC PK SC; do withlib $l line vt_$l --mode train --steps 4 --warmup 2; done
  for l in PK SC PK SC; do withlib $l line rt16_$l --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3; done
  echo "== 3x3 BatchNorm folding (opt-in of round 4)"
  for r in a b; do
    DREAM_BN_FUSION_3X3=0 line rt16_bn3off_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_BN_FUSION_3X3=1 line rt16_bn3on_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  ;;
prio)
  # weight-gradient side stream at the lowest / highest / normal HIP stream priority (resnet_h training), + the tests this round touched
  echo "== pytest (touched)"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 -k "structured or bench_dp_check or graph_replay_equals_eager or two_physical or resnet_h_train_step or batchnorm_in_the_3x3" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  for r in a b; do for pr in default low high; do
    DREAM_SIDE_STREAM_PRIORITY=$pr line rt16_${pr}_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done; done
  for pr in default low; do
    DREAM_SIDE_STREAM_PRIORITY=$pr DREAM_OVERLAP_MAX_FRAMES=128 line rt128_overlap_${pr} --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  done
  line rt128_inorder --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  ;;
prio2)
  # is the b=128 gain of the low-priority weight-gradient stream stable?  (round 4: forced overlap at 128 frames scattered 311..450)
  for r in a b c; do
    DREAM_SIDE_STREAM_PRIORITY=low DREAM_OVERLAP_MAX_FRAMES=128 line rt128_overlap_low_$r --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
    line rt128_inorder_$r --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  done
  for r in a b; do
    DREAM_SIDE_STREAM_PRIORITY=low DREAM_VGG_OVERLAP_MAX_FRAMES=128 line vt_overlap_low_$r --mode train --steps 3 --warmup 2
    line vt_inorder_$r --mode train --steps 3 --warmup 2
  done
  DREAM_SIDE_STREAM_PRIORITY=low DREAM_OVERLAP_MAX_FRAMES=128 line rt64_overlap_low --arch resnet_h --mode train --batch 64 --steps 4 --warmup 2
  DREAM_SIDE_STREAM_PRIORITY=low line rt64_default_low --arch resnet_h --mode train --batch 64 --steps 4 --warmup 2
  line rt64_default --arch resnet_h --mode train --batch 64 --steps 4 --warmup 2
  timeout 600 python tools/microbench_wino4_mask.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/microbench_wino4_mask.txt
  ;;
mask2)
  # after the epilogue fix (residual / mask loads grouped one column ahead): parity, the microbench again, the training lines
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "winograd4 or wino4 or structured or backward_ops or train_step or train_steps or reference_golden or variant or skip or convT or conv4x4 or general_conv" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  timeout 600 python tools/microbench_wino4_mask.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/microbench_wino4_mask.txt
  line vt_a --mode train --steps 4 --warmup 2
  line vt_b --mode train --steps 4 --warmup 2
  line rt16_a --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rt128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  line vft32 --arch vgg_f --mode train --batch 32 --steps 4 --warmup 2
  line dflt
  ;;
g1)
  # gemm1x1 epilogue: residual / mask operands one row ahead in every form (product) vs the predicated per-row loop (build/libG0.so)
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "conv1x1 or resnet or bn_fused or stress" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  for r in a b c; do
    line rt16_pipe_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    withlib G0 line rt16_G0_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  line rh128_pipe --arch resnet_h --batch 128
  withlib G0 line rh128_G0 --arch resnet_h --batch 128
  line rh16_pipe --arch resnet_h --batch 16
  withlib G0 line rh16_G0 --arch resnet_h --batch 16
  line rt128_pipe --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  withlib G0 line rt128_G0 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  ;;
pool)
  # the pooled tensor stored by the training conv's own launch (MODE 4) against the stand-alone max-pool pass
  echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "pool_both or pool_in_the_training or train_step or train_steps or reference_golden or variant or data_parallel" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  for r in a b c; do
    DREAM_POOL_IN_TRAINING_CONV=1 line vt_fused_$r --mode train --steps 4 --warmup 2
    DREAM_POOL_IN_TRAINING_CONV=0 line vt_separate_$r --mode train --steps 4 --warmup 2
  done
  line rt16 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rt128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  ;;
sbatch)
  for r in a b; do for n in 1 4 8 16; do
    DREAM_SIDE_BATCH=$n line rt16_batch${n}_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done; done
  for n in 1 8; do DREAM_SIDE_BATCH=$n line rt128_batch${n} --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2; done
  echo "== pytest"; DREAM_SIDE_BATCH=8 timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "resnet_h_train_step or resnet_f_train_step or vgg_f_train or reference_golden" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  ;;
waux)
  for r in a b c; do for l in W0 W1 W16; do withlib $l line dflt_${l}_$r; done; done
  for l in W0 W16 W0 W16; do withlib $l line vt_$l --mode train --steps 4 --warmup 2; done
  ;;
cold)
  # first bench on a cold box: the forced overlap at 128 frames with record_stream() (arg rec) or kept references (arg keep)
  K=0; [ "$2" = keep ] && K=1
  DREAM_SIDE_KEEP=$K DREAM_OVERLAP_MAX_FRAMES=128 line rt128_$2_cold --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  DREAM_SIDE_KEEP=$K DREAM_OVERLAP_MAX_FRAMES=128 line rt128_$2_second --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  ;;
keep)
  # the weight-gradient stream's inputs kept referenced until the join (DREAM_SIDE_KEEP=1) against record_stream(): is the 300-frames/s mode
  # of the forced overlap at 128 frames the allocator?
  for r in a b c d; do
    DREAM_SIDE_KEEP=1 DREAM_OVERLAP_MAX_FRAMES=128 line rt128_keep_$r --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
    DREAM_SIDE_KEEP=0 DREAM_OVERLAP_MAX_FRAMES=128 line rt128_rec_$r --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  done
  line rt128_inorder --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  for r in a b c; do
    DREAM_SIDE_KEEP=1 line rt16_keep_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_SIDE_KEEP=0 line rt16_rec_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  ;;
diag)
  DREAM_W4_DIAG_KS=${2:-128,256,15,2,16,8} timeout 600 python tools/wino4_diag.py run --batch 128 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/wino4_diag.txt
  ;;
tg)
  # the one-device training step as two hipGraph replays (DREAM_TRAIN_GRAPH=1) against the eager step, on the round's last tree
  for r in a b; do
    DREAM_TRAIN_GRAPH=0 line rt16_eager_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
    DREAM_TRAIN_GRAPH=1 line rt16_graph_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  done
  DREAM_TRAIN_GRAPH=0 line vt_eager --mode train --steps 4 --warmup 3
  DREAM_TRAIN_GRAPH=1 line vt_graph --mode train --steps 4 --warmup 3
  ;;
tg2)
  # which of the round's changes slowed the captured training step down?
  DREAM_TRAIN_GRAPH=1 line g_all --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  DREAM_TRAIN_GRAPH=1 DREAM_SIDE_STREAM_PRIORITY=default line g_prio_default --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  DREAM_TRAIN_GRAPH=1 DREAM_SIDE_KEEP=0 line g_keep0 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  DREAM_TRAIN_GRAPH=1 DREAM_SIDE_STREAM_PRIORITY=default DREAM_SIDE_KEEP=0 line g_prio_default_keep0 --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  DREAM_TRAIN_GRAPH=1 DREAM_BN_FUSION_3X3=0 line g_bn3off --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  DREAM_TRAIN_GRAPH=1 DREAM_OVERLAP_WGRAD=0 line g_nooverlap --arch resnet_h --mode train --batch 16 --steps 10 --warmup 4
  ;;
tg3)
  # the runtime's side of a captured training step: how many queues the graph executor spreads the two branches over
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  line eager_a $R
  DREAM_TRAIN_GRAPH=1 line g_a $R
  for q in 1 2 8; do DREAM_TRAIN_GRAPH=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=$q line g_q$q $R; done
  for b in 1 100000; do DREAM_TRAIN_GRAPH=1 DEBUG_HIP_GRAPH_BATCH_SIZE=$b line g_b$b $R; done
  DREAM_TRAIN_GRAPH=1 GPU_MAX_HW_QUEUES=8 line g_hw8 $R
  DREAM_TRAIN_GRAPH=1 line g_b $R
  line eager_b $R
  DREAM_TRAIN_GRAPH=1 AMD_LOG_LEVEL=4 timeout 240 python bench.py --arch resnet_h --mode train --batch 16 --steps 1 --warmup 3 --no-cpu-baseline --no-secondary 2>&1 \
    | grep -E "max_streams|parallel streams|max streams" | sed 's/^.*\(\[hipGraph\]\|GraphExec\)/\1/' | sort | uniq -c | tee $O/graph_streams.txt
  ;;
tg4)
  # the captured backward as a SEQUENCE of graphs, the weight-gradient leaves replayed on a live second stream (DREAM_TRAIN_GRAPH_SPLIT=leaves per segment)
  export LINE_TIMEOUT=150
  echo "== pytest"; timeout 400 python -m pytest tests -m gpu -q -x --timeout 150 -k "one_device_training_step" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  line eager_a $R
  DREAM_TRAIN_GRAPH=1 line g_a $R
  for n in 8 2 4 16 32 1000 8; do DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=$n line g_split$n $R; done
  line eager_b $R
  ;;
tg5)
  # tg4 again: the leaf segments on the eager steps' own second stream (live) or on a fresh one (new); repeats for the spread
  export LINE_TIMEOUT=150
  echo "== pytest"; timeout 400 python -m pytest tests -m gpu -q -x --timeout 150 -k "one_device_training_step" > $O/pytest.log 2>&1; echo "rc=$?"; tail -1 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  line eager_a $R
  for r in a b c; do
    DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=8 line live8_$r $R
    DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=8 DREAM_TRAIN_GRAPH_SPLIT_STREAM=new line new8_$r $R
  done
  DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=4 line live4 $R
  line eager_b $R
  ;;
tg6)
  # split graphs as the default of the one-device captured step: every test that captures, and the lines
  export LINE_TIMEOUT=150
  echo "== pytest"; timeout 500 python -m pytest tests -m gpu -q -x --timeout 200 -k "graph or data_parallel or replica or dp_" > $O/pytest.log 2>&1; echo "rc=$?"; tail -1 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  V="--mode train --batch 16 --steps 10 --warmup 4"
  line rt16_eager $R
  DREAM_TRAIN_GRAPH=1 line rt16_graph $R
  DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=0 line rt16_graph_one $R
  line vt16_eager $V
  DREAM_TRAIN_GRAPH=1 line vt16_graph $V
  DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=0 line vt16_graph_one $V
  ;;
tg7)
  # split graphs: the last segments tapered down to single leaves (DREAM_TRAIN_GRAPH_SPLIT_TAPER=1) or equal segments (=0)
  export LINE_TIMEOUT=150
  echo "== pytest"; timeout 400 python -m pytest tests -m gpu -q -x --timeout 150 -k "one_device_training_step or graph_replay_equals" > $O/pytest.log 2>&1; echo "rc=$?"; tail -1 $O/pytest.log
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  line eager_a $R
  for r in a b; do
    DREAM_TRAIN_GRAPH=1 line taper_$r $R
    DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT_TAPER=0 line equal_$r $R
  done
  DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=12 line taper12 $R
  DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=5 line taper5 $R
  line eager_b $R
  ;;
suite)
  # the whole GPU suite + smoke() on the last tree, then leaves per segment of the split graphs (8 / 12) against eager
  echo "== pytest gpu (all)"; timeout 450 python -m pytest tests -m gpu -q --timeout 200 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
  echo "== smoke"; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-60,100-190
  export LINE_TIMEOUT=60
  R="--arch resnet_h --mode train --batch 16 --steps 10 --warmup 4"
  line eager_a $R
  for r in a b; do
    DREAM_TRAIN_GRAPH=1 line split8_$r $R
    DREAM_TRAIN_GRAPH=1 DREAM_TRAIN_GRAPH_SPLIT=12 line split12_$r $R
  done
  line eager_b $R
  ;;
tg8)
  # the captured step where the second stream is NOT used (resnet_h at 128 frames, vgg_q at 128): the plan is one main graph
  export LINE_TIMEOUT=80
  DREAM_TRAIN_GRAPH=1 line rt128_graph --arch resnet_h --mode train --batch 128 --steps 4 --warmup 3
  DREAM_TRAIN_GRAPH=1 line vt_graph --mode train --steps 4 --warmup 3
  ;;
tgprof)
  # dispatch timeline of the captured step (split graphs): do the leaf segments run beside the main segments?
  R="$PWD"
  (cd /tmp && DREAM_TRAIN_GRAPH=1 timeout 110 rocprofv3 --kernel-trace --stats -d "$R/$O/prof" -o g -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 5 --warmup 4 --no-cpu-baseline --no-secondary > "$R/$O/rocprof.log" 2>&1); echo "rc=$?"
  grep -h '^{"metric' $O/rocprof.log | cut -c1-160
  db=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" $O/bench_resnet_h_train16_graph adam_kernel 3 > $O/summ.log 2>&1; echo "summary rc=$?"; rm -rf $O/prof
  cat $O/bench_resnet_h_train16_graph_concurrency.txt
  ;;
sgrid)
  # convT on four-wavefront workgroups where the eight-wavefront grid would leave CUs empty (DREAM_WINO_SMALL_GRID=1, default) vs always eight (=0)
  echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "conv_transpose4x4_winograd or resnet_h_train_step or structured" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for r in a b c; do
    DREAM_WINO_SMALL_GRID=1 line rt16_nw4_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_WINO_SMALL_GRID=0 line rt16_nw8_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  for r in a b; do
    DREAM_WINO_SMALL_GRID=1 line rf32_nw4_$r --arch resnet_f --batch 32
    DREAM_WINO_SMALL_GRID=0 line rf32_nw8_$r --arch resnet_f --batch 32
    DREAM_WINO_SMALL_GRID=1 line rh16_nw4_$r --arch resnet_h --batch 16
    DREAM_WINO_SMALL_GRID=0 line rh16_nw8_$r --arch resnet_h --batch 16
  done
  ;;
mask)
  timeout 600 python tools/microbench_wino4_mask.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/microbench_wino4_mask.txt
  ;;
stagger)
  timeout 600 python tools/ab_wino4_stagger.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/ab_wino4_stagger.txt
  ;;
dflt)
  # the default line alone, into the final set's directory (after a change that touches nothing measured there)
  O=gpurun_out/r05_final; mkdir -p $O
  T0=$SECONDS; timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; tail -1 $O/bench_default.log | cut -c1-2500; echo "$((SECONDS - T0)) s wall"
  ;;
final)
  # The round's measurement set (one GPU call): full GPU suite, smoke(), rehearsals of both multi-GPU paths on this one GPU (2 and 8 ranks /
  # replicas), rocprofv3 kernel tables, the calibrated HBM-traffic and SQ PMC passes, ResNet training traffic + per-kernel counters of the 1x1
  # GEMMs inside the step, the secondary lines, ten consecutive runs of the ResNet step (the round-5 "slow mode"), layer profiles,
  # micro-benchmarks, and the default line LAST (it picks up this bench.py's PMC traffic).  `collect` (dev container) copies the summaries
  # into profiles/r06_*.
  R="$PWD"
  summ() { db=$(ls $1/*.db $1/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" "$2" $3 $4 > $O/summ.log 2>&1; echo "summary $2 rc=$?"; rm -rf "$1"; }
  lraw() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | cut -c1-160; }
  if [ "$2" != "notests" ]; then
    echo "== pytest gpu (all)"; timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
    echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-60,100-190
  fi
  echo "== rehearsals on this one GPU (the numbers mean nothing)"
  DREAM_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 2 --warmup 1 --secondary-steps 2 --secondary-train-steps 3 > $O/rehearsal_8ranks_gloo.log 2>&1; echo "rc=$?"; tail -1 $O/rehearsal_8ranks_gloo.log | cut -c1-60,100-190
  DREAM_BENCH_GPU_IDS=0,0,0,0,0,0,0,0 timeout 900 python bench.py --gpus 8 --single-process --arch resnet_h --mode train --steps 4 --warmup 3 --global-batch 128 --no-cpu-baseline > $O/rehearsal_single_process_8replicas_train.log 2>&1; echo "rc=$?"; tail -1 $O/rehearsal_single_process_8replicas_train.log | cut -c1-60,100-190
  DREAM_BENCH_GPU_IDS=0,0,0,0 timeout 600 python bench.py --gpus 4 --single-process --arch resnet_h --mode train --steps 6 --warmup 3 --global-batch 64 --no-cpu-baseline > $O/rehearsal_single_process_4replicas_train.log 2>&1; echo "rc=$?"; tail -1 $O/rehearsal_single_process_4replicas_train.log | cut -c1-60,100-190
  echo "== rocprof default bench (kernel trace)"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_default" -o dflt -- python "$R/bench.py" --no-cpu-baseline --no-secondary > "$R/$O/rocprof_default.log" 2>&1); echo "rc=$?"; grep -h '^{"metric' $O/rocprof_default.log | cut -c1-60,100-190
  summ $O/prof_default $O/bench_default peaks_kernel 2
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $C"; (cd /tmp && DREAM_BENCH_PMC_CALIBRATE=1 timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_$C" -o pmc -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > "$R/$O/pmc_$C.log" 2>&1); echo "rc=$?"
  done
  python tools/pmc_traffic.py $(ls $O/pmc_FETCH_SIZE/*/*.db $O/pmc_FETCH_SIZE/*.db 2>/dev/null | head -1) $(ls $O/pmc_WRITE_SIZE/*/*.db $O/pmc_WRITE_SIZE/*.db 2>/dev/null | head -1) $O/pmc_traffic.json | grep -i "ratio\|raw_to" | head
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
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_rt_$C" -o pmc -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > "$R/$O/pmc_rt_$C.log" 2>&1); echo "resnet train pmc $C rc=$?"
  done
  python tools/pmc_traffic.py $(ls $O/pmc_rt_FETCH_SIZE/*/*.db $O/pmc_rt_FETCH_SIZE/*.db 2>/dev/null | head -1) $(ls $O/pmc_rt_WRITE_SIZE/*/*.db $O/pmc_rt_WRITE_SIZE/*.db 2>/dev/null | head -1) $O/pmc_traffic_resnet_train.json "bn_,gemm1x1_kernel,wgrad1x1,conv_mfma_kernel,conv_wino_kernel,conv_wino_stat_kernel,conv_wino4_kernel,wgrad_kernel<,wgrad_wino,adam,pack" --arch resnet_h --mode train --batch 16 --steps 1 --warmup 1 | head -40
  rm -rf $O/pmc_rt_FETCH_SIZE $O/pmc_rt_WRITE_SIZE
  # per-kernel counters of the 1x1 GEMM family inside the training step (four passes; the PMC passes serialise the dispatches)
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  P2="SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SALU"
  P3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
  P4="TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_EA0_RDREQ_sum"
  for pass in 1 2 3 4; do
    eval C=\$P$pass
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$R/$O/pmc_g_$pass" -o p -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > "$R/$O/pmc_g_$pass.log" 2>&1); echo "pmc gemm pass $pass rc=$?"
  done
  python tools/pmc_kernels.py gemm1x1_kernel,wgrad1x1_kernel,wgrad_wino_lds_kernel $O/pmc_g_* > $O/pmc_gemm1x1_in_step.txt 2> $O/pmc_g.err; head -12 $O/pmc_gemm1x1_in_step.txt | cut -c1-260
  rm -rf $O/pmc_g_[1-9]
  lraw train --mode train --steps 5 --warmup 2
  echo "== ten consecutive 10-step runs of the ResNet-101 training step at 16 frames"
  for i in 0 1 2 3 4 5 6 7 8 9; do lraw resnet_h_train16_run$i --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3; done
  cp $O/bench_resnet_h_train16_run9.log $O/bench_resnet_h_train16.log
  lraw resnet_h_train128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  lraw resnet_f_b32 --arch resnet_f --batch 32
  lraw resnet_h_b128 --arch resnet_h --batch 128
  lraw vgg_f_b32 --arch vgg_f --batch 32
  lraw resnet_f_train32 --arch resnet_f --mode train --batch 32 --steps 4 --warmup 2
  echo "== rocprof train"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_train" -o train -- python "$R/bench.py" --mode train --steps 3 --warmup 2 --no-cpu-baseline > "$R/$O/rocprof_train.log" 2>&1); echo "rc=$?"
  summ $O/prof_train $O/bench_train adam_kernel 1
  echo "== rocprof resnet_h train16"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 8 --warmup 3 --no-cpu-baseline > "$R/$O/rocprof_rtrain.log" 2>&1); echo "rc=$?"
  summ $O/prof_rtrain $O/bench_resnet_h_train16 adam_kernel 3
  echo "== layer profiles"
  for cfg in "resnet_h train 16" "vgg_q infer 128" "vgg_q train 128" "resnet_f infer 32"; do set -- $cfg
    timeout 300 python tools/layer_profile.py --arch $1 --mode $2 --batch $3 --top 45 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_$1_$2$3.txt; head -3 $O/layer_profile_$1_$2$3.txt | cut -c1-60,100-190
  done
  echo "== microbenches"; timeout 300 python tools/microbench_wino4.py --batch 128 2>&1 | grep -v amdgpu.ids > $O/microbench_wino4_b128.txt; tail -1 $O/microbench_wino4_b128.txt
  timeout 300 python tools/microbench_gemm_forms.py --shapes layer3 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/microbench_gemm_forms.txt; tail -3 $O/microbench_gemm_forms.txt
  timeout 300 python tools/microbench_conv1x1.py --batch 16 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/microbench_conv1x1_b16.txt; tail -1 $O/microbench_conv1x1_b16.txt
  timeout 400 python tools/wgw_diag.py run --batch 128 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/wgw_diag.txt; tail -4 $O/wgw_diag.txt | cut -c1-300
  cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json
  cp $O/pmc_traffic_resnet_train.json profiles/r06_pmc_traffic_resnet_train.json
  echo "== default bench (with the PMC traffic of this bench.py)"; T0=$SECONDS; timeout 1200 python bench.py > $O/bench_default.log 2> $O/bench_default.err; tail -1 $O/bench_default.log | cut -c1-2400; echo "$((SECONDS - T0)) s wall"
  du -sh $O
  ;;
collect)
  # dev container: gpurun_out/r06_final -> profiles/r06_*
  O=gpurun_out/r06_final
  for n in default train resnet_h_train16; do
    cp $O/bench_${n}_kernel_stats.csv profiles/r06_bench_${n}_kernel_stats.csv
    cp $O/bench_${n}_conv_dispatches.csv profiles/r06_bench_${n}_conv_dispatches.csv
    [ -f $O/bench_${n}_concurrency.txt ] && cp $O/bench_${n}_concurrency.txt profiles/r06_bench_${n}_concurrency.txt
  done
  cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json
  cp $O/pmc_traffic_resnet_train.json profiles/r06_pmc_traffic_resnet_train.json
  cp $O/pmc_mfma.json profiles/r06_pmc_mfma.json
  cp $O/pmc_gemm1x1_in_step.txt profiles/r06_pmc_gemm1x1_in_step.txt
  for n in default train resnet_h_train16 resnet_h_train128 resnet_f_b32 resnet_h_b128 vgg_f_b32 resnet_f_train32; do tail -1 $O/bench_$n.log > profiles/r06_bench_${n}_line.json; done
  python - <<'PY' > profiles/r06_resnet_h_train16_ten_runs.txt
import json
v = []
for i in range(10):
    d = json.loads(open("gpurun_out/r06_final/bench_resnet_h_train16_run%d.log" % i).read().strip().split("\n")[-1])
    v.append(d["value"])
m = sorted(v)[len(v) // 2]
print("# ten consecutive `bench.py --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3` runs on one box (tools/gpu_round.sh final): frames/s")
print(" ".join("%.1f" % x for x in v))
print("median %.1f  min %.1f (%.1f %%)  max %.1f (+%.1f %%)" % (m, min(v), 100 * (min(v) / m - 1), max(v), 100 * (max(v) / m - 1)))
PY
  cat profiles/r06_resnet_h_train16_ten_runs.txt
  grep -h '^{"metric' $O/rocprof_default.log > profiles/r06_bench_default_under_rocprof_line.json
  grep -E "passed|failed" $O/pytest_gpu.log | tail -1 > profiles/r06_pytest_gpu_tail.txt
  for f in $O/layer_profile_*.txt; do cp $f profiles/r06_$(basename $f); done
  cp $O/microbench_wino4_b128.txt profiles/r06_microbench_wino4_b128.txt
  cp $O/microbench_gemm_forms.txt profiles/r06_microbench_gemm_forms_final.txt
  cp $O/microbench_conv1x1_b16.txt profiles/r06_microbench_conv1x1_b16.txt
  cp $O/wgw_diag.txt profiles/r06_wgw_diag.txt
  tail -1 $O/rehearsal_8ranks_gloo.log > profiles/r06_rehearsal_8ranks_gloo_line.json
  tail -1 $O/rehearsal_single_process_8replicas_train.log > profiles/r06_rehearsal_single_process_8replicas_train_line.json
  tail -1 $O/rehearsal_single_process_4replicas_train.log > profiles/r06_rehearsal_single_process_4replicas_train_line.json
  tail -3 $O/smoke.log > profiles/r06_smoke_tail.txt
  ls profiles/r06_*
  ;;
esac
