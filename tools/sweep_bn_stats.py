#!/usr/bin/env python
"""Stand-alone BatchNorm statistics (csrc/bn.hip, bn_stats_slab_kernel): pixels per partial row (= workgroups along the pixel axis,
hence levels of the ticket tree) on ResNet-101's tensor shapes at 16 frames per GPU; forward statistics and the backward reductions,
apply kernels beside them.  Round-robin, ten back-to-back launches per sample.  python tools/sweep_bn_stats.py [--batch 16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import _hip, ops  # noqa: E402

SHAPES = [(100, 64), (100, 256), (50, 128), (50, 512), (25, 256), (25, 1024), (13, 512), (13, 2048), (208, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    pxs = [64, 128, 256, 640, 1024]
    for res, c in SHAPES:
        z = torch.randn(a.batch, res, res, c, device="cuda")
        dy = torch.randn_like(z)
        bn = torch.nn.BatchNorm2d(c).cuda()
        ctr = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
        ab, mean, invstd = ops.bn_stats(z, bn, ctr)
        y = ops.bn_apply_ab(z, ab, None, True)
        best = {}
        for _ in range(a.reps):
            for px in pxs:
                _hip.call("dream_bn_stats_set_pixels_per_row", px)
                for name, fn in (("fwd", lambda: ops.bn_stats(z, bn, ctr)), ("bwd", lambda: ops.bn_bwd_stats(z, dy, mean, invstd, ctr, y_act=y))):
                    fn()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(10):
                        fn()
                    e.record()
                    torch.cuda.synchronize()
                    best[(px, name)] = min(best.get((px, name), 1e9), s.elapsed_time(e) / 10)
        _hip.call("dream_bn_stats_set_pixels_per_row", 128)
        mb = z.numel() * 4 / 1e6
        print("%4dx%-4d C=%-5d %6.1f MB  " % (res, res, c, mb) + "  ".join("px%-4d fwd %5.1f bwd %5.1f us" % (px, best[(px, "fwd")] * 1e3, best[(px, "bwd")] * 1e3) for px in pxs), flush=True)


if __name__ == "__main__":
    main()
