#!/usr/bin/env python
"""Run one conv layer shape repeatedly on one kernel (for rocprofv3 --pmc passes).
usage: one_layer.py <fp32|f16x3|wino|wino4> <variant> <res> <cin> <cout> <batch> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dream_amd import _hip, ops
kind, variant, res, cin, cout, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
lib = _hip.lib()
x = torch.randn(batch, res, res, cin, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
bias = torch.randn(cout, device="cuda")
if kind == "wgwino":
    dy = torch.randn(batch, res, res, cout, device="cuda")
    for _ in range(reps):
        ops.conv3x3_wgrad_winograd(x, dy, cout, cin, want_bias=False)
elif kind == "wino4":
    u, _ = ops.pack_weight_winograd4(w, 0)
    for _ in range(reps):
        ops.conv3x3_winograd4(x, u, cout, None, bias, None, 1)
elif kind == "wino":
    u, _ = ops.pack_weight_winograd(w, 0)
    lib.dream_conv3x3_winograd_set_variant(variant)
    for _ in range(reps):
        ops.conv3x3_winograd(x, u, cout, None, bias, None, 1)
elif kind == "fp32":
    packed, rows, _, _ = ops.pack_weight(w, 0)
    lib.dream_conv3x3_set_variant(variant)
    for _ in range(reps):
        ops.conv3x3(x, packed, bias, cout, 1)
else:
    p16 = ops.pack_conv_weight_f16x3(w, 0)
    amax = ops.absmax(x)
    lib.dream_conv_f16x3_set_variant(variant)
    for _ in range(reps):
        ops.conv2d_f16x3(x, amax, p16, cout, 3, None, bias, None, 1)
torch.cuda.synchronize()
