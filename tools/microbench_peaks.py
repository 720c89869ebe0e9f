#!/usr/bin/env python
"""Belief-map peak extraction (csrc/peaks.hip) alone, HIP events: the keypoint rule of DreamNetwork.inference (column pass + fused row
pass / scan + finish) and the two Gaussian passes, on the map shapes of the BASELINE configurations.  GB/s = the bytes the form has to
move (keypoint rule: the maps once + the column pass' result written and read; Gaussian: read + write per pass) per second.
Usage: python tools/microbench_peaks.py [--reps 7]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import ops  # noqa: E402

SHAPES = [(32, 17, 416, 416, "resnet_f b=32"), (128, 7, 100, 100, "vgg_q b=128"), (16, 7, 208, 208, "resnet_h b=16")]


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
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(5)
    for b, k, h, w, name in SHAPES:
        m = torch.rand((b, k, h, w), device="cuda", generator=g) * 0.02
        m[:, :, h // 3, w // 2] = 0.9
        nbytes = m.numel() * 4
        ms_rule = timeit(lambda: ops.keypoints_from_belief_maps(m, 0.5), args.reps)
        ms_gauss = timeit(lambda: ops.gaussian_sigma3(m.reshape(b * k, h, w)), args.reps)
        print("%-14s %3d x %2d maps of %3d x %3d  keypoint rule %7.3f ms (%5.2f TB/s over 3 x the maps)   two Gaussian passes %7.3f ms "
              "(%5.2f TB/s over 4 x the maps)" % (name, b, k, h, w, ms_rule, 3 * nbytes / ms_rule / 1e9, ms_gauss, 4 * nbytes / ms_gauss / 1e9))


if __name__ == "__main__":
    main()
