#!/usr/bin/env python
"""K-split of the LDS-free 1x1 GEMM (csrc/gemm1x1.hip): 1 / 2 / 4 waves per wave tile, plain and with the BatchNorm statistics in
the epilogue, on ResNet-101's 1x1 layer shapes -- is the host heuristic (by tile count) the fastest choice?  Round-robin timing.
python tools/sweep_conv1x1_ks.py [--batch 16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import _hip, ops  # noqa: E402

LAYERS = [(100, 64, 256), (100, 256, 64), (50, 128, 512), (50, 512, 128), (25, 256, 1024), (25, 1024, 256), (13, 512, 2048), (13, 2048, 512)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    for res, cin, cout in LAYERS:
        x = torch.randn(a.batch, res, res, cin, device="cuda")
        w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
        packed, rows = ops.pack_conv1x1_weight(w, 0)
        bn = torch.nn.BatchNorm2d(cout).cuda()
        ctr = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
        kss = [k for k in (0, 1, 2, 4) if k == 0 or cin % (32 * k) == 0]
        best = {(k, v): 1e9 for k in kss for v in ("plain", "bn")}
        for _ in range(a.reps):
            for k in kss:
                _hip.call("dream_conv1x1_set_ksplit", k)
                for v, fn in (("plain", lambda: ops.conv1x1(x, packed, rows)), ("bn", lambda: ops.conv1x1_bn(x, packed, rows, bn, ctr))):
                    fn()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(10):
                        fn()
                    e.record()
                    torch.cuda.synchronize()
                    best[(k, v)] = min(best[(k, v)], s.elapsed_time(e) / 10)
        _hip.call("dream_conv1x1_set_ksplit", 0)
        fl = 2.0 * a.batch * res * res * cin * cout
        print("%4d %5d->%5d  " % (res, cin, cout) + "  ".join("ks%d plain %6.1f us bn %6.1f us" % (k, best[(k, "plain")] * 1e3, best[(k, "bn")] * 1e3) for k in kss)
              + "   | heuristic: %.1f TF plain, %.1f TF with statistics" % (fl / best[(0, "plain")] / 1e9, fl / best[(0, "bn")] / 1e9), flush=True)


if __name__ == "__main__":
    main()
