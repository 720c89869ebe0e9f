#!/usr/bin/env python
"""Interleaved A/B of conv variants at the bench batch size (thermal drift hits both arms equally)."""
import argparse, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dream_amd import _hip, ops

LAYERS = [(400, 64, 64, 1), (200, 64, 128, 1), (200, 128, 128, 1), (100, 128, 256, 1), (100, 256, 256, 1),
          (50, 256, 512, 1), (50, 512, 512, 1), (25, 512, 512, 1), (100, 128, 64, 0), (100, 64, 64, 1)]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    lib = _hip.lib()
    for (res, cin, cout, flags) in LAYERS:
        x = torch.randn(args.batch, res, res, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        bias = torch.randn(cout, device="cuda")
        packed, rows, _, _ = ops.pack_weight(w, 0)
        arms = [0, 3] if cout > 64 else [1, 4]
        flops = 2.0 * args.batch * res * res * cin * cout * 9
        times = {v: [] for v in arms}
        for r in range(args.rounds + 1):
            for v in arms:
                lib.dream_conv3x3_set_variant(v)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ops.conv3x3(x, packed, bias, cout, flags); e.record(); torch.cuda.synchronize()
                if r > 0: times[v].append(s.elapsed_time(e))
        lib.dream_conv3x3_set_variant(-1)
        print("%4d %3d->%3d | " % (res, cin, cout) + "  ".join("%s: %.3f ms (%.1f TF)" % (lib.dream_conv3x3_variant_name(v).decode(), statistics.median(t), flops / statistics.median(t) / 1e9) for v, t in times.items()), flush=True)
        del x

def main16():
    """f16x3 arms: 256x64 tile vs 8-wave 256x128 vs the s_setprio builds of both."""
    lib = _hip.lib()
    for (res, cin, cout, flags) in LAYERS:
        x = torch.randn(128, res, res, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        bias = torch.randn(cout, device="cuda")
        p16 = ops.pack_conv_weight_f16x3(w, 0)
        amax = ops.absmax(x)
        arms = [1, 6, 4, 7] if cout > 64 else [1, 6, 5]
        flops = 2.0 * 128 * res * res * cin * cout * 9
        times = {v: [] for v in arms}
        for r in range(6):
            for v in arms:
                lib.dream_conv_f16x3_set_variant(v)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ops.conv2d_f16x3(x, amax, p16, cout, 3, None, bias, None, flags); e.record(); torch.cuda.synchronize()
                if r > 0: times[v].append(s.elapsed_time(e))
        lib.dream_conv_f16x3_set_variant(-1)
        print("f16x3 %4d %3d->%3d | " % (res, cin, cout) + "  ".join("v%d: %.3f ms (%.0f TF)" % (v, statistics.median(t), flops / statistics.median(t) / 1e9) for v, t in times.items()), flush=True)
        del x


if __name__ == "__main__":
    if "--f16x3" in sys.argv:
        sys.argv.remove("--f16x3")
        main16()
    else:
        main()
