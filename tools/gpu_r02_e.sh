#!/bin/bash
# Round 2, call E: re-run of the tests touched since call D, ResNet training with the Winograd 3x3 convs, secondary configs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02e
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest (subset)"; timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -s -k "winograd or ragged or data_parallel or resnet or structured or variants" > $O/pytest.log 2>&1; echo "rc=$?"; tail -2 $O/pytest.log; grep "^FAILED\|^E " $O/pytest.log | head
line() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$n', round(d['value'],1),'fps', round(d['ms_per_step'],2),'ms frac',round(r['frac'],3),'exec',round(r['executed_frac'],3),'share',round(r['share_of_step_time'],2), 'split', round(d.get('split_precision',{}).get('value',0),1))"; }
line resnet_h_train16 --arch resnet_h --mode train --batch 16 --steps 5 --warmup 2
line resnet_h_train16_direct --arch resnet_h --mode train --batch 16 --steps 5 --warmup 2 --conv-algorithm direct
line resnet_h_train128 --arch resnet_h --mode train --batch 128 --steps 3 --warmup 1
line resnet_f_b32 --arch resnet_f --batch 32
line resnet_h_b128 --arch resnet_h --batch 128
line vgg_f_b32 --arch vgg_f --batch 32
line vgg_f_train32 --arch vgg_f --mode train --batch 32 --steps 3 --warmup 1
line vgg_q_train --mode train --steps 4 --warmup 1
