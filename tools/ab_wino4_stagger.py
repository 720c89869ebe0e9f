#!/usr/bin/env python
"""F(4x4,3x3) kernel: start-up stagger of the persistent workgroups (csrc/conv_wino4.hip, Wino4Params::stagger_n) on the vgg_q layer
shapes, round-robin over the settings on one box; same bits (asserted).   python tools/ab_wino4_stagger.py [--batch 128] [--reps 6]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import _hip, ops  # noqa: E402

# (res, cin, cout, calls per vgg_q forward, fused pool?)
LAYERS = [(400, 64, 64, 1, 1), (200, 64, 128, 1, 0), (200, 128, 128, 1, 1), (100, 128, 256, 1, 0), (100, 256, 256, 3, 0), (50, 256, 512, 1, 0),
          (50, 512, 512, 3, 0), (25, 512, 512, 4, 0), (50, 256, 256, 1, 0)]
SETTINGS = [(0, 0), (16, 12), (16, 25), (16, 50), (16, 100), (64, 50), (4, 25)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--settings", default="")
    a = ap.parse_args()
    settings = [tuple(int(v) for v in s.split(":")) for s in a.settings.split(",")] if a.settings else SETTINGS
    tot = [0.0] * len(settings)
    for res, cin, cout, count, pool in LAYERS:
        x = torch.randn(a.batch, res, res, cin, device="cuda").relu_()
        w = (torch.rand(cout, cin, 3, 3, device="cuda") * 2 - 1) * (6.0 / (9 * cin)) ** 0.5
        bias = torch.randn(cout, device="cuda") * 0.05
        u4, _ = ops.pack_weight_winograd4(w, 0)
        flags = ops.CONV_RELU | (ops.CONV_POOL2 if pool else 0)
        ref, best = None, [1e9] * len(settings)
        for n, pct in settings:
            _hip.call("dream_conv3x3_winograd4_set_stagger", n, pct)
            y = ops.conv3x3_winograd4(x, u4, cout, None, bias, None, flags)
            if ref is None:
                ref = y
            assert torch.equal(ref, y)
        for _ in range(a.reps):
            for i, (n, pct) in enumerate(settings):
                _hip.call("dream_conv3x3_winograd4_set_stagger", n, pct)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.conv3x3_winograd4(x, u4, cout, None, bias, None, flags)
                e.record()
                torch.cuda.synchronize()
                best[i] = min(best[i], s.elapsed_time(e))
        fl = 2.0 * a.batch * res * res * cin * cout * 9 / 4.0
        print("%4d %4d->%4d x%d  " % (res, cin, cout, count) + "  ".join("%d:%d %.3f ms (%.3f)" % (n, pct, t, fl / t / 1e9 / 157.3)
                                                                          for (n, pct), t in zip(settings, best)), flush=True)
        for i, t in enumerate(best):
            tot[i] += count * t
        del x, ref, y
    _hip.call("dream_conv3x3_winograd4_set_stagger", -1, 0)
    print("sum over a vgg_q forward pass: " + "  ".join("%d:%d %.2f ms" % (n, pct, t) for (n, pct), t in zip(settings, tot)))


if __name__ == "__main__":
    main()
