#!/usr/bin/env python
"""Winograd F(4x4,3x3) (csrc/conv_wino4.hip) vs F(2x2,3x3) (csrc/conv_wino.hip) on the stride-1 3x3 layer shapes of DREAM-vgg-Q with
>= 64 output channels (HIP events, interleaved A/B on one box), each with its error against an fp64 direct convolution of a
sub-batch.  TFLOP/s are DIRECT-algorithm FLOPs per second.  Usage: python tools/microbench_wino4.py [--batch 128] [--reps 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from dream_amd import ops  # noqa: E402

LAYERS = [  # (res, cin, cout, fused pool?, count in vgg_q)
    (400, 64, 64, 1, 1),        # the kernel's narrow workgroup shape
    (200, 64, 128, 0, 1), (200, 128, 128, 1, 1), (100, 128, 256, 0, 1), (100, 256, 256, 0, 3),
    (50, 256, 512, 0, 1), (50, 512, 512, 0, 3), (25, 512, 512, 0, 4), (50, 256, 256, 0, 1),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rows, t2, t4 = [], 0.0, 0.0
    for (res, cin, cout, pool, count) in LAYERS:
        b = args.batch
        x = torch.randn(b, res, res, cin, device="cuda").relu_()             # post-ReLU activations, as in the network
        w = (torch.rand(cout, cin, 3, 3, device="cuda") * 2 - 1) * (6.0 / (9 * cin)) ** 0.5
        bias = torch.randn(cout, device="cuda") * 0.05
        u2, _ = ops.pack_weight_winograd(w, 0)
        u4, _ = ops.pack_weight_winograd4(w, 0)
        flags = ops.CONV_RELU | (ops.CONV_POOL2 if pool else 0)
        y2 = ops.conv3x3_winograd(x, u2, cout, None, bias, None, flags)
        y4 = ops.conv3x3_winograd4(x, u4, cout, None, bias, None, flags)
        # fp64 truth on the first image (no ReLU / pool: the raw conv), error relative to the output maximum
        r2 = ops.conv3x3_winograd(x[:1].contiguous(), u2, cout, None, bias, None, 0)
        r4 = ops.conv3x3_winograd4(x[:1].contiguous(), u4, cout, None, bias, None, 0)
        ref = F.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
        e2 = float((r2.double() - ref).abs().max() / ref.abs().max())
        e4 = float((r4.double() - ref).abs().max() / ref.abs().max())
        ms2 = timeit(lambda: ops.conv3x3_winograd(x, u2, cout, None, bias, None, flags), args.reps)
        ms4 = timeit(lambda: ops.conv3x3_winograd4(x, u4, cout, None, bias, None, flags), args.reps)
        flops = 2.0 * b * res * res * cin * cout * 9
        rows.append({"layer": [res, cin, cout, pool], "count": count, "f2_ms": ms2, "f4_ms": ms4, "f2_tflops_direct_equiv": flops / ms2 / 1e9,
                     "f4_tflops_direct_equiv": flops / ms4 / 1e9, "f4_mfma_frac_of_peak": flops / 4.0 / ms4 / 1e9 / 157.3,
                     "f2_err_vs_fp64": e2, "f4_err_vs_fp64": e4, "f4_vs_f2_maxdiff": float((y2 - y4).abs().max())})
        t2 += count * ms2
        t4 += count * ms4
        print("%4d %4d->%4d pool%d x%d  F(2x2) %7.3f ms %6.1f TF | F(4x4) %7.3f ms %6.1f TF-equiv (%.2f of peak on its own MACs) "
              "speedup %.2f | err vs fp64 / max: F2 %.1e F4 %.1e" % (res, cin, cout, pool, count, ms2, flops / ms2 / 1e9, ms4, flops / ms4 / 1e9,
                                                                   flops / 4.0 / ms4 / 1e9 / 157.3, ms2 / ms4, e2, e4), flush=True)
        del x, y2, y4
    print("sum over these vgg_q layers (b=%d): F(2x2) %.2f ms, F(4x4) %.2f ms, speedup %.2f" % (args.batch, t2, t4, t2 / t4))
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"batch": args.batch, "layers": rows, "f2_ms": t2, "f4_ms": t4}, f, indent=1)


if __name__ == "__main__":
    main()
