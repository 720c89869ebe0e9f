#!/bin/bash
# Round-4 development call (kept as ONE parameterised script: tools/gpu_r04_step.sh <stage>), not part of the measurement set.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_$1
mkdir -p $O
export TMPDIR=/tmp
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | cut -c1-230; }
case "$1" in
bn1)
  echo "== pytest (new + touched)"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "bn_fused or resnet_h_train_step or reference_golden or resnet_training_ops or data_parallel or conv1x1" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  for r in 1 2; do
    DREAM_BN_FUSION=1 line rt16_fused_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_BN_FUSION=0 line rt16_three_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  DREAM_BN_FUSION=1 line rt128_fused --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
  DREAM_BN_FUSION=0 line rt128_three --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
  timeout 300 python tools/layer_profile.py --arch resnet_h --mode train --batch 16 --top 50 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_fused.txt; head -40 $O/layer_profile_fused.txt | cut -c1-160
  ;;
w4pin)
  for r in 1 2; do
    DREAM_BN_FUSION=1 line rt16_fused_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_BN_FUSION=0 line rt16_three_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  timeout 300 python tools/layer_profile.py --arch resnet_h --mode train --batch 16 --top 30 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_fused.txt; sed -n 2,22p $O/layer_profile_fused.txt | cut -c1-120
  timeout 300 python tools/ab_wino4_pinning.py 2>&1 | grep -v amdgpu.ids | tee $O/ab_wino4_pinning.txt
  DREAM_W4_YMAP=0 line dflt_walked
  DREAM_W4_YMAP=1 line dflt_pinned
  DREAM_W4_YMAP=0 line dflt_walked2
  DREAM_W4_YMAP=1 line dflt_pinned2
  ;;
bn3)
  echo "== pytest (new + touched)"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "bn_fused or resnet_h_train_step or reference_golden or resnet_training_ops or data_parallel or conv1x1 or skip_connections or variant" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
  for r in 1 2; do
    DREAM_BN_FUSION=1 line rt16_fused_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_BN_FUSION=0 line rt16_three_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  DREAM_BN_FUSION=1 line rt128_fused --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
  DREAM_BN_FUSION=0 line rt128_three --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
  timeout 300 python tools/layer_profile.py --arch resnet_h --mode train --batch 16 --top 30 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|imagenet_init" > $O/layer_profile_fused.txt; sed -n 2,24p $O/layer_profile_fused.txt | cut -c1-120
  ;;
full)
  echo "== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
  echo "== default bench"; /usr/bin/time -f "%e s wall" timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; tail -1 $O/bench_default.log | cut -c1-1500; tail -1 $O/bench_default.err
  ;;
ks)
  timeout 300 python tools/sweep_conv1x1_ks.py 2>&1 | grep -v amdgpu.ids | tee $O/sweep_conv1x1_ks_b16.txt
  timeout 300 python tools/sweep_conv1x1_ks.py --batch 128 --reps 3 2>&1 | grep -v amdgpu.ids | tee $O/sweep_conv1x1_ks_b128.txt
  ;;
midbar)
  echo "== pytest (F(4x4) kernels, fixtures)"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "winograd4 or wino4 or structured or convT4 or golden or pinned" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  DREAM_W4_DIAG_KS=2101 timeout 300 python tools/wino4_diag.py run --batch 128 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/wino4_midbarrier.txt
  line dflt_a
  line dflt_b
  ;;
bnrows)
  timeout 300 python tools/sweep_bn_stats.py 2>&1 | grep -v amdgpu.ids | tee $O/sweep_bn_stats_b16.txt
  ;;
bnfull)
  timeout 900 python -m pytest tests -m gpu -q -x -s --timeout 600 -k "batchnorm_paths_agree" 2>&1 | grep -v "Warning\|warn" | tail -6
  ;;
conc)
  summ() { db=$(ls $1/*.db $1/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py "$db" "$2" "$3" "$4" > $O/summ.log 2>&1; echo "summary $2 rc=$?"; tail -9 $O/summ.log; rm -rf "$1"; }
  R="$PWD"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 5 --warmup 3 --no-cpu-baseline > "$R/$O/rocprof_rtrain.log" 2>&1); echo "rc=$?"
  summ $O/prof_rtrain $O/bench_resnet_h_train16 adam_kernel 2
  (cd /tmp && DREAM_BN_FUSION=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_rtrain3" -o rtrain -- python "$R/bench.py" --arch resnet_h --mode train --batch 16 --steps 5 --warmup 3 --no-cpu-baseline > "$R/$O/rocprof_rtrain3.log" 2>&1); echo "rc=$?"
  summ $O/prof_rtrain3 $O/bench_resnet_h_train16_three_launch_bn adam_kernel 2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_vtrain" -o vtrain -- python "$R/bench.py" --mode train --steps 3 --warmup 2 --no-cpu-baseline > "$R/$O/rocprof_vtrain.log" 2>&1); echo "rc=$?"
  summ $O/prof_vtrain $O/bench_train adam_kernel 1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_dflt" -o dflt -- python "$R/bench.py" --no-cpu-baseline --no-secondary > "$R/$O/rocprof_dflt.log" 2>&1); echo "rc=$?"
  summ $O/prof_dflt $O/bench_default peaks_kernel 2
  ;;
mc)
  echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "resnet or train or adam or optimizer or abi or general_conv or variant" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  line rt16_a --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rt16_b --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rh128 --arch resnet_h --batch 128
  line rf32 --arch resnet_f --batch 32
  line rt128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
  ;;
mp)
  echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "resnet or repack or batched or training_ops or data_parallel" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  line rt16_a --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rt16_b --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rt128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
  line rft16 --arch resnet_f --mode train --batch 16 --steps 5 --warmup 3
  ;;
vt)
  echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "backward_ops or train_step or train_steps or vgg or variant or first" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  line vt_a --mode train --steps 4 --warmup 2
  line vt_b --mode train --steps 4 --warmup 2
  line vft --arch vgg_f --mode train --batch 32 --steps 4 --warmup 2
  ;;
b3)
  # first measurement of round 5: the 3x3 convs' BatchNorm inside the Winograd kernel (opt-in) against the default, alternating
  echo "== pytest"; timeout 600 python -m pytest tests -m gpu -q -x --timeout 400 -k "conv3x3_bn_fused_ops or batchnorm_in_the_3x3 or resnet_h_train_step" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  for r in a b; do
    DREAM_BN_FUSION_3X3=0 line rt16_off_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
    DREAM_BN_FUSION_3X3=1 line rt16_on_$r --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  done
  DREAM_BN_FUSION_3X3=0 line rt128_off --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  DREAM_BN_FUSION_3X3=1 line rt128_on --arch resnet_h --mode train --batch 128 --steps 3 --warmup 2
  ;;
tg)
  echo "== pytest"; timeout 600 python -m pytest tests -m gpu -q -x -s --timeout 400 -k "one_device_training_step_as_graph or graph_replay_equals_eager" 2>&1 | grep -v "Warning\|warn" | tail -6
  line rt16_eager_a --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  DREAM_TRAIN_GRAPH=1 line rt16_graph_a --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rt16_eager_b --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  DREAM_TRAIN_GRAPH=1 line rt16_graph_b --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rt128_eager --arch resnet_h --mode train --batch 128 --steps 3 --warmup 3
  DREAM_TRAIN_GRAPH=1 line rt128_graph --arch resnet_h --mode train --batch 128 --steps 3 --warmup 3
  line vt_eager --mode train --steps 4 --warmup 3
  DREAM_TRAIN_GRAPH=1 line vt_graph --mode train --steps 4 --warmup 3
  ;;
wb)
  echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "wgrad or backward_ops or train_step or train_steps or reference_golden or variant or general_conv or convT" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log
  DREAM_WGRAD_BIAS_FUSION=0 line vt_sum_a --mode train --steps 4 --warmup 2
  DREAM_WGRAD_BIAS_FUSION=1 line vt_fused_a --mode train --steps 4 --warmup 2
  DREAM_WGRAD_BIAS_FUSION=0 line vt_sum_b --mode train --steps 4 --warmup 2
  DREAM_WGRAD_BIAS_FUSION=1 line vt_fused_b --mode train --steps 4 --warmup 2
  line rt16_a --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  line rt16_b --arch resnet_h --mode train --batch 16 --steps 10 --warmup 3
  ;;
esac
