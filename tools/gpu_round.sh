#!/bin/bash
# ONE parameterised GPU-box script (replaces the per-round gpu_rNN_* families): tools/gpu_round.sh <stage> [args].
# Run through gpurun from the repo root; writes under gpurun_out/r05_<stage>/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_$1
mkdir -p $O
export TMPDIR=/tmp
PROD=dream_amd/libdream_hip.so
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-secondary > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
    print('$n', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms', 'frac', round(r.get('frac',0),4), 'exec', r.get('executed_frac'))
except Exception as e: print('$n', 'FAILED', e)
"; }
withlib() { cp $PROD /tmp/lib_keep.so; cp build/lib$1.so $PROD; shift; "$@"; cp /tmp/lib_keep.so $PROD; }
case "$1" in
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
stagger)
  timeout 600 python tools/ab_wino4_stagger.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/ab_wino4_stagger.txt
  ;;
esac
