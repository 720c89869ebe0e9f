#!/usr/bin/env python
"""Winograd F(2x2,3x3) kernel vs the direct MFMA kernel on the stride-1 3x3 layer shapes of DREAM-vgg-Q (HIP events,
interleaved A/B on one box).  TFLOP/s are DIRECT-algorithm FLOPs per second (so > 157 means beyond the fp32 roof).
Usage: python tools/microbench_wino.py [--batch 128] [--reps 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import _hip, ops  # noqa: E402

LAYERS = [  # (res, cin, cout, fused pool?, count in vgg_q)
    (400, 64, 64, 1, 1), (200, 64, 128, 0, 1), (200, 128, 128, 1, 1), (100, 128, 256, 0, 1), (100, 256, 256, 0, 3),
    (50, 256, 512, 0, 1), (50, 512, 512, 0, 3), (25, 512, 512, 0, 4), (50, 256, 256, 0, 1), (100, 128, 64, 0, 1),
    (100, 64, 64, 0, 1), (100, 64, 32, 0, 1),
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
    ap.add_argument("--variant", type=int, default=0, help="0: workgroup width by layer, 4 / 8: forced")
    args = ap.parse_args()
    _hip.lib().dream_conv3x3_winograd_set_variant(args.variant)
    print("winograd variant", args.variant)
    rows, t_dir, t_win = [], 0.0, 0.0
    for (res, cin, cout, pool, count) in LAYERS:
        b = args.batch
        x = torch.randn(b, res, res, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
        bias = torch.randn(cout, device="cuda")
        packed, prow, _, _ = ops.pack_weight(w, 0)
        u, _ = ops.pack_weight_winograd(w, 0)
        flags = ops.CONV_RELU | (ops.CONV_POOL2 if pool else 0)
        yd = ops.conv3x3(x, packed, bias, cout, flags)
        yw = ops.conv3x3_winograd(x, u, cout, None, bias, None, flags)
        diff = float((yd - yw).abs().max()) / max(1.0, float(yd.abs().max()))
        ms_d = timeit(lambda: ops.conv3x3(x, packed, bias, cout, flags), args.reps)
        ms_w = timeit(lambda: ops.conv3x3_winograd(x, u, cout, None, bias, None, flags), args.reps)
        flops = 2.0 * b * res * res * cin * cout * 9
        rows.append({"layer": [res, cin, cout, pool], "count": count, "direct_ms": ms_d, "wino_ms": ms_w,
                     "direct_tflops": flops / ms_d / 1e9, "wino_tflops_direct_equiv": flops / ms_w / 1e9,
                     "wino_mfma_frac_of_peak": flops / 2.25 / ms_w / 1e9 / 157.3, "rel_diff": diff})
        t_dir += count * ms_d
        t_win += count * ms_w
        print("%4d %4d->%4d pool%d x%d  direct %7.3f ms %6.1f TF | wino %7.3f ms %6.1f TF-equiv (%.2f of peak on its own MACs) "
              "speedup %.2f  rel diff %.1e" % (res, cin, cout, pool, count, ms_d, flops / ms_d / 1e9, ms_w, flops / ms_w / 1e9,
                                               flops / 2.25 / ms_w / 1e9 / 157.3, ms_d / ms_w, diff), flush=True)
        del x, yd, yw
    print("sum over the vgg_q layers (b=%d): direct %.2f ms, winograd %.2f ms, speedup %.2f" % (args.batch, t_dir, t_win, t_dir / t_win))
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"batch": args.batch, "layers": rows, "direct_ms": t_dir, "wino_ms": t_win}, f, indent=1)


if __name__ == "__main__":
    main()
