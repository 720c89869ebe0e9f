#!/usr/bin/env python
"""Weight gradient of the decoder's ConvTranspose2d(4,2,1) layers (dream/models.py:37-136): the nine-position F(2x2,2x2) form of
round 6 (csrc/wgrad_wino.hip, CONVT: weight + bias gradient in one launch + a reduction) against the direct kernel it replaces
(+ the stand-alone bias pass), layer by layer, interleaved on one box.    python tools/microbench_convT_wgrad.py [--batch 16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import ops  # noqa: E402

LAYERS = [(13, 2048, 256), (26, 256, 256), (52, 256, 256), (104, 256, 256), (208, 256, 256)]     # the last one: resnet_f's upsample2


def timeit(fn, reps=5, inner=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / inner)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--digest", action="store_true", help="seeded inputs; print a sha256 of the F(2x2,2x2) form's outputs per layer "
                    "(bit-identity of two library builds: tools/gpu_round.sh g6n)")
    a = ap.parse_args()
    if a.digest:
        import hashlib
        for (res, cin, cout) in LAYERS[:a.layers]:
            g = torch.Generator(device="cuda").manual_seed(res)
            x = torch.randn(a.batch, res, res, cin, device="cuda", generator=g)
            dy = torch.randn(a.batch, 2 * res, 2 * res, cout, device="cuda", generator=g)
            w, b = ops.convT4x4_wgrad_winograd(x, dy)
            print("%4d^2 %5d->%4d  sha256(dw) %s  sha256(db) %s" % (res, cin, cout, hashlib.sha256(w.cpu().numpy().tobytes()).hexdigest()[:16],
                                                                hashlib.sha256(b.cpu().numpy().tobytes()).hexdigest()[:16]))
        return
    td = tw = 0.0
    for (res, cin, cout) in LAYERS[:a.layers]:
        x = torch.randn(a.batch, res, res, cin, device="cuda")
        dy = torch.randn(a.batch, 2 * res, 2 * res, cout, device="cuda")
        d = ops.convT4x4_wgrad(x, dy)
        w, b = ops.convT4x4_wgrad_winograd(x, dy)
        scale = float(d.abs().max())
        diff = float((d - w).abs().max()) / scale
        bdiff = float((b - ops.channel_sum(dy)).abs().max()) / float(b.abs().max())

        def direct():
            ops.convT4x4_wgrad(x, dy)
            ops.channel_sum(dy)
        ms_d = timeit(direct)
        ms_w = timeit(lambda: ops.convT4x4_wgrad_winograd(x, dy))
        flops = 2.0 * a.batch * res * res * cin * cout * 16
        td += ms_d
        tw += ms_w
        print("%4d^2 %5d->%4d  direct + bias pass %8.1f us %6.1f TF | F(2x2,2x2) %8.1f us %6.1f TF direct-equivalent (%.2f of the MFMA peak on "
              "its own multiplications)  speed-up %.2f  rel diff %.1e bias %.1e" % (
                  res, cin, cout, ms_d * 1e3, flops / ms_d / 1e9, ms_w * 1e3, flops / ms_w / 1e9, flops * 9 / 16 / ms_w / 1e9 / 157.3,
                  ms_d / ms_w, diff, bdiff), flush=True)
    print("sum over the decoder layers (b=%d): direct %.2f ms, F(2x2,2x2) %.2f ms, speed-up %.2f" % (a.batch, td, tw, td / tw))


if __name__ == "__main__":
    main()
