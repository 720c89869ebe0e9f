#!/usr/bin/env python
"""ConvTranspose2d(k4,s2,p1) of the ResNet decoder: minimal filtering on the Winograd kernel (9 multiplications per 2x2 outputs
of a phase) vs the direct sub-pixel kernel (16), interleaved on one box.  TFLOP/s are DIRECT-algorithm FLOPs per second.
python tools/microbench_convT.py [--batch 32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import ops  # noqa: E402

LAYERS = [(13, 2048, 256), (26, 256, 256), (52, 256, 256), (104, 256, 256), (208, 256, 256)]


def timeit(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 2)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    td = tw = tbd = tbw = 0.0
    for (res, cin, cout) in LAYERS:
        if res == 208 and a.batch > 32:
            continue
        x = torch.randn(a.batch, res, res, cin, device="cuda")
        wT = torch.randn(cin, cout, 4, 4, device="cuda") * 0.03
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        pd, rows = ops.pack_convT4x4_weight(wT)
        u4, _ = ops.pack_convT4x4_winograd_weight(wT)
        yd = ops.conv_transpose4x4s2(x, pd, cout, sc, sh, ops.CONV_RELU)
        yw = ops.conv_transpose4x4s2_winograd(x, u4, cout, sc, sh, ops.CONV_RELU)
        diff = float((yd - yw).abs().max()) / float(yd.abs().max())
        ms_d = timeit(lambda: ops.conv_transpose4x4s2(x, pd, cout, sc, sh, ops.CONV_RELU))
        ms_w = timeit(lambda: ops.conv_transpose4x4s2_winograd(x, u4, cout, sc, sh, ops.CONV_RELU))
        dy = torch.randn(a.batch, 2 * res, 2 * res, cout, device="cuda")
        pb, rows_b = ops.pack_convT4x4_bwd_weight(wT)
        u4b, _ = ops.pack_convT4x4_winograd_weight(wT, 1)
        gd = ops.conv4x4s2(dy, pb, rows_b)
        gw = ops.conv4x4s2_winograd(dy, u4b, cin)
        bdiff = float((gd[..., :cin] - gw).abs().max()) / float(gd.abs().max())
        ms_bd = timeit(lambda: ops.conv4x4s2(dy, pb, rows_b))
        ms_bw = timeit(lambda: ops.conv4x4s2_winograd(dy, u4b, cin))
        tbd += ms_bd
        tbw += ms_bw
        flops = 2.0 * a.batch * res * res * cin * cout * 16
        bline = " || data gradient: direct %8.3f ms | winograd %8.3f ms speedup %.2f rel diff %.1e" % (ms_bd, ms_bw, ms_bd / ms_bw, bdiff)
        del dy, gd, gw
        td += ms_d
        tw += ms_w
        print("%4d -> %4d  %5d->%4d  direct %8.3f ms %6.1f TF | winograd %8.3f ms %6.1f TF-equiv (%.2f of peak on its own MACs) "
              "speedup %.2f  rel diff %.1e" % (res, 2 * res, cin, cout, ms_d, flops / ms_d / 1e9, ms_w, flops / ms_w / 1e9,
                                               flops * 9 / 16 / ms_w / 1e9 / 157.3, ms_d / ms_w, diff) + bline, flush=True)
        del x, yd, yw
    print("sum over the decoder layers (b=%d): direct %.2f ms, winograd %.2f ms, speedup %.2f; data gradient: direct %.2f ms, "
          "winograd %.2f ms, speedup %.2f" % (a.batch, td, tw, td / tw, tbd, tbw, tbd / tbw))


if __name__ == "__main__":
    main()
