#!/usr/bin/env python
"""Winograd-domain weight gradient vs the direct MFMA weight-gradient kernel on the vgg_q 3x3 layer shapes (HIP events,
interleaved on one box).  TFLOP/s are DIRECT-algorithm FLOPs per second.  Usage: python tools/microbench_wgrad_wino.py [--batch 128]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import _hip, ops  # noqa: E402

LAYERS = [(400, 64, 64, 1), (200, 64, 128, 1), (200, 128, 128, 1), (100, 128, 256, 1), (100, 256, 256, 3), (50, 256, 512, 1),
          (50, 512, 512, 3), (25, 512, 512, 4), (50, 256, 256, 1), (100, 128, 64, 1), (100, 64, 64, 1), (100, 64, 32, 1)]


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
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--digest", action="store_true", help="seeded inputs; sha256 of the Winograd form's dw / db per layer (bit-identity of two "
                    "library builds), no timing")
    ap.add_argument("--only-winograd", action="store_true", help="time the Winograd form only")
    args = ap.parse_args()
    if args.digest:
        import hashlib
        for (res, cin, cout, count) in LAYERS:
            g = torch.Generator(device="cuda").manual_seed(res + cin)
            x = torch.randn(args.batch, res, res, cin, device="cuda", generator=g)
            dy = torch.randn(args.batch, res, res, cout, device="cuda", generator=g)
            w, db = ops.conv3x3_wgrad_winograd(x, dy, cout, cin)
            print("%4d %4d->%4d  sha256(dw) %s  sha256(db) %s" % (res, cin, cout, hashlib.sha256(w.cpu().numpy().tobytes()).hexdigest()[:16],
                  "-" if db is None else hashlib.sha256(db.cpu().numpy().tobytes()).hexdigest()[:16]))
            del x, dy, w
        return
    if args.only_winograd:
        tw = 0.0
        for (res, cin, cout, count) in LAYERS:
            x = torch.randn(args.batch, res, res, cin, device="cuda")
            dy = torch.randn(args.batch, res, res, cout, device="cuda")
            ms_w = timeit(lambda: ops.conv3x3_wgrad_winograd(x, dy, cout, cin), args.reps)
            flops = 2.0 * args.batch * res * res * cin * cout * 9
            tw += count * ms_w
            print("%4d %4d->%4d x%d  winograd %8.3f ms %6.1f TF-equiv (%.3f of peak on its own MACs)" % (
                res, cin, cout, count, ms_w, flops / ms_w / 1e9, flops / 2.25 / ms_w / 1e9 / 157.3), flush=True)
            del x, dy
        print("sum over the vgg_q layers (b=%d): winograd %.2f ms" % (args.batch, tw))
        return
    td = tw = 0.0
    for (res, cin, cout, count) in LAYERS:
        b = args.batch
        x = torch.randn(b, res, res, cin, device="cuda")
        dy = torch.randn(b, res, res, cout, device="cuda")
        a, _ = ops.conv3x3_wgrad(x, dy, cout, cin)
        w, _ = ops.conv3x3_wgrad_winograd(x, dy, cout, cin)
        diff = float((a - w).abs().max()) / float(a.abs().max())
        ms_d = timeit(lambda: ops.conv3x3_wgrad(x, dy, cout, cin), args.reps)
        ms_w = timeit(lambda: ops.conv3x3_wgrad_winograd(x, dy, cout, cin), args.reps)
        _hip.lib().dream_conv3x3_wgrad_winograd_set_version(1)          # the register-only kernel, for comparison
        ms_w1 = timeit(lambda: ops.conv3x3_wgrad_winograd(x, dy, cout, cin), args.reps)
        _hip.lib().dream_conv3x3_wgrad_winograd_set_version(0)
        flops = 2.0 * b * res * res * cin * cout * 9
        td += count * ms_d
        tw += count * ms_w
        print("%4d %4d->%4d x%d  direct %8.3f ms %6.1f TF | winograd %8.3f ms %6.1f TF-equiv (%.2f of peak on its own MACs) "
              "speedup %.2f  rel diff %.1e | register-only version %8.3f ms" % (
                  res, cin, cout, count, ms_d, flops / ms_d / 1e9, ms_w, flops / ms_w / 1e9,
                  flops / 2.25 / ms_w / 1e9 / 157.3, ms_d / ms_w, diff, ms_w1), flush=True)
        del x, dy, a, w
    print("sum over the vgg_q layers (b=%d): direct %.2f ms, winograd %.2f ms, speedup %.2f" % (args.batch, td, tw, td / tw))


if __name__ == "__main__":
    main()
