#!/usr/bin/env python
"""1x1 convs of ResNet-101: the LDS-free GEMM kernel vs the direct conv kernel, forward and data-gradient operators, at the
per-GPU batch of configs[3] (16) and at 128.  python tools/microbench_conv1x1.py [--batch 16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import ops  # noqa: E402

LAYERS = [(100, 64, 256, 3), (100, 256, 64, 3), (50, 128, 512, 4), (50, 512, 128, 4), (25, 256, 1024, 23), (25, 1024, 256, 23),
          (13, 512, 2048, 3), (13, 2048, 512, 3), (100, 64, 64, 1)]


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    td = tg = twd = twg = 0.0
    for (res, cin, cout, count) in LAYERS:
        x = torch.randn(a.batch, res, res, cin, device="cuda")
        w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
        res_t = torch.randn(a.batch, res, res, cout, device="cuda")
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        pd, rows, _ = ops.pack_conv_weight(w, 0)
        pg, _ = ops.pack_conv1x1_weight(w, 0)
        yd = ops.conv2d(x, pd, cout, 1, 1, sc, sh, res_t, ops.CONV_RELU)
        yg = ops.conv1x1(x, pg, cout, sc, sh, res_t, ops.CONV_RELU)
        diff = float((yd - yg).abs().max()) / float(yd.abs().max())
        ms_d = timeit(lambda: ops.conv2d(x, pd, cout, 1, 1, sc, sh, res_t, ops.CONV_RELU))
        ms_g = timeit(lambda: ops.conv1x1(x, pg, cout, sc, sh, res_t, ops.CONV_RELU))
        dy = torch.randn(a.batch, res, res, cout, device="cuda")
        wd = ops.conv2d_wgrad(x, dy, cout, cin, 1, 1)[0]
        wg = ops.conv1x1_wgrad(x, dy, cout, cin)
        wdiff = float((wd - wg).abs().max()) / float(wd.abs().max())
        ms_wd = timeit(lambda: ops.conv2d_wgrad(x, dy, cout, cin, 1, 1))
        ms_wg = timeit(lambda: ops.conv1x1_wgrad(x, dy, cout, cin))
        twd += count * ms_wd
        twg += count * ms_wg
        flops = 2.0 * a.batch * res * res * cin * cout
        wline = " || wgrad direct %7.1f us %5.1f TF | gemm %7.1f us %5.1f TF speedup %.2f rel diff %.1e" % (
            ms_wd * 1e3, flops / ms_wd / 1e9, ms_wg * 1e3, flops / ms_wg / 1e9, ms_wd / ms_wg, wdiff)
        td += count * ms_d
        tg += count * ms_g
        print("%4d %5d->%5d x%-2d direct %7.1f us %6.1f TF | gemm %7.1f us %6.1f TF (%.2f of peak) speedup %.2f  rel diff %.1e" % (
            res, cin, cout, count, ms_d * 1e3, flops / ms_d / 1e9, ms_g * 1e3, flops / ms_g / 1e9, flops / ms_g / 1e9 / 157.3,
            ms_d / ms_g, diff) + wline, flush=True)
    print("sum over the ResNet-101 1x1 layers (b=%d): direct %.2f ms, gemm %.2f ms, speedup %.2f; weight gradients: direct %.2f ms, "
          "gemm %.2f ms, speedup %.2f" % (a.batch, td, tg, td / tg, twd, twg, twd / twg))


if __name__ == "__main__":
    main()
