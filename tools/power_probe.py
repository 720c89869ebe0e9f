#!/usr/bin/env python
"""Is a kernel limited by its instruction stream or by the chip's power management?  The same launches on all-zero operands
(no bit toggles in the matrix cores: the clock stays up) and on random operands.  python tools/power_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import ops  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 4)
    return best


def main():
    b = 128
    for (res, cin, cout) in [(100, 256, 256), (50, 512, 512), (400, 64, 64)]:
        line = []
        for kind in ("randn", "zeros", "small ints"):
            def make(*shape):
                if kind == "randn":
                    return torch.randn(*shape, device="cuda")
                if kind == "zeros":
                    return torch.zeros(*shape, device="cuda")
                return torch.randint(0, 2, shape, device="cuda").float()
            x = make(b, res, res, cin)
            dy = make(b, res, res, cout)
            w = make(cout, cin, 3, 3)
            u, rows = ops.pack_weight_winograd(w, 0)
            flops = 2.0 * b * res * res * cin * cout * 9
            t_f = timeit(lambda: ops.conv3x3_winograd(x, u, rows, None, None, None, 1))
            t_w = timeit(lambda: ops.conv3x3_wgrad_winograd(x, dy, cout, cin))
            t_d = timeit(lambda: ops.conv3x3_wgrad(x, dy, cout, cin))
            line.append("%s: wino fwd %.3f ms (%.2f) wino wgrad %.3f ms (%.2f) direct wgrad %.3f ms (%.2f)" % (
                kind, t_f, flops / 2.25 / t_f / 1e9 / 157.3, t_w, flops / 2.25 / t_w / 1e9 / 157.3, t_d, flops / t_d / 1e9 / 157.3))
            del x, dy, w, u
        print("%d %d->%d b=%d | " % (res, cin, cout, b) + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
