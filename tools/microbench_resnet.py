#!/usr/bin/env python
"""Per-layer timing of the ResNet-101 trunk / decoder conv shapes for every selectable variant of conv_mfma_kernel.
Usage (GPU box): python tools/microbench_resnet.py --batch 16"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dream_amd import _hip, ops  # noqa: E402

# (H=W, cin, cout, k, stride)
SHAPES = [(100, 64, 64, 1, 1), (100, 64, 64, 3, 1), (100, 64, 256, 1, 1), (100, 256, 64, 1, 1),
          (50, 128, 128, 3, 1), (50, 128, 512, 1, 1), (50, 512, 128, 1, 1),
          (25, 256, 256, 3, 1), (25, 256, 1024, 1, 1), (25, 1024, 256, 1, 1),
          (13, 512, 512, 3, 1), (13, 512, 2048, 1, 1), (13, 2048, 512, 1, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    lib = _hip.lib()
    nv = lib.dream_conv3x3_num_variants()
    dev = torch.device("cuda")
    for (h, cin, cout, k, stride) in SHAPES:
        x = torch.randn(args.batch, h, h, cin, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        packed, rows, _ = ops.pack_conv_weight(w, 0)
        flops = 2.0 * args.batch * h * h * cin * cout * k * k
        out = []
        for v in range(-1, nv):
            lib.dream_conv3x3_set_variant(v)
            try:
                ops.conv2d(x, packed, rows, k, stride)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(args.reps):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    ops.conv2d(x, packed, rows, k, stride)
                    e.record()
                    e.synchronize()
                    best = min(best, s.elapsed_time(e))
                out.append("%s:%5.1f" % ("h" if v < 0 else str(v), flops / best / 1e9))
            except Exception as err:
                out.append("%s:  err" % v)
            finally:
                lib.dream_conv3x3_set_variant(-1)
        print("%3d %4d->%4d k%d | %s" % (h, cin, cout, k, " ".join(out)), flush=True)


if __name__ == "__main__":
    main()
