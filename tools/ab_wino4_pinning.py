#!/usr/bin/env python
"""F(4x4,3x3) kernel, output-channel blocks pinned to XCDs vs walked by every XCD (csrc/conv_wino4.hip, Wino4Params::ymap), on
the vgg_q layer shapes with more than 128 output channels, interleaved A/B on one box; same bits either way (asserted).
Usage: python tools/ab_wino4_pinning.py [--batch 128] [--reps 7]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import _hip, ops  # noqa: E402

LAYERS = [(100, 128, 256, 1), (100, 256, 256, 3), (50, 256, 512, 1), (50, 512, 512, 3), (25, 512, 512, 4), (50, 256, 256, 1), (50, 512, 256, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    tot = [0.0, 0.0]
    for res, cin, cout, count in LAYERS:
        x = torch.randn(a.batch, res, res, cin, device="cuda").relu_()
        w = (torch.rand(cout, cin, 3, 3, device="cuda") * 2 - 1) * (6.0 / (9 * cin)) ** 0.5
        bias = torch.randn(cout, device="cuda") * 0.05
        u4, _ = ops.pack_weight_winograd4(w, 0)
        outs, best = [], [1e9, 1e9]
        for pin in (0, 1):
            _hip.call("dream_conv3x3_winograd4_set_channel_block_pinning", pin)
            outs.append(ops.conv3x3_winograd4(x, u4, cout, None, bias, None, ops.CONV_RELU))
        assert torch.equal(outs[0], outs[1])
        for _ in range(a.reps):
            for pin in (0, 1):
                _hip.call("dream_conv3x3_winograd4_set_channel_block_pinning", pin)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.conv3x3_winograd4(x, u4, cout, None, bias, None, ops.CONV_RELU)
                e.record()
                torch.cuda.synchronize()
                best[pin] = min(best[pin], s.elapsed_time(e))
        fl = 2.0 * a.batch * res * res * cin * cout * 9 / 4.0
        print("%4d %4d->%4d x%d  walked %7.3f ms (%.3f of peak)  pinned %7.3f ms (%.3f)  pinned/walked %.3f" % (
            res, cin, cout, count, best[0], fl / best[0] / 1e9 / 157.3, best[1], fl / best[1] / 1e9 / 157.3, best[1] / best[0]), flush=True)
        tot[0] += count * best[0]
        tot[1] += count * best[1]
        del x, outs
    _hip.call("dream_conv3x3_winograd4_set_channel_block_pinning", -1)
    print("sum: walked %.2f ms, pinned %.2f ms" % tuple(tot))


if __name__ == "__main__":
    main()
