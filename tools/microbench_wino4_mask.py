#!/usr/bin/env python
"""F(4x4,3x3) kernel: what the ReLU-mask epilogue of the data-gradient launches (MODE 3: one mask load per output) costs against the plain
epilogue (MODE 0) and the residual-add one (MODE 2), same layer, round-robin on one box.   python tools/microbench_wino4_mask.py [--batch 128]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import ops  # noqa: E402

LAYERS = [(400, 64, 64), (200, 128, 128), (100, 256, 256), (50, 512, 512), (25, 512, 512), (200, 128, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    for res, cin, cout in LAYERS:
        x = torch.randn(a.batch, res, res, cin, device="cuda")
        w = (torch.rand(cout, cin, 3, 3, device="cuda") * 2 - 1) * (6.0 / (9 * cin)) ** 0.5
        u4, _ = ops.pack_weight_winograd4(w, 0)
        mask = torch.randn(a.batch, res, res, cout, device="cuda")
        modes = {"plain": (None, 0), "relu mask": (mask, ops.CONV_RELUMASK), "residual add": (mask, 0)}
        best = {k: 1e9 for k in modes}
        for k, (r, f) in modes.items():
            ops.conv3x3_winograd4(x, u4, cout, None, None, r, f)
        for _ in range(a.reps):
            for k, (r, f) in modes.items():
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.conv3x3_winograd4(x, u4, cout, None, None, r, f)
                e.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], s.elapsed_time(e))
        fl = 2.0 * a.batch * res * res * cin * cout * 9 / 4.0
        print("%4d %4d->%4d  " % (res, cin, cout) + "  ".join("%s %.3f ms (%.3f)" % (k, t, fl / t / 1e9 / 157.3) for k, t in best.items()), flush=True)
        del x, mask


if __name__ == "__main__":
    main()
