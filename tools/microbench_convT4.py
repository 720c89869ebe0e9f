#!/usr/bin/env python
"""ConvTranspose2d(k4,s2,p1) 256 -> 256 by minimal filtering: the F(2x2,3x3) kernel with its 9-position phase patterns against the
F(4x4,3x3) kernel with its 25-position ones (round-robin, HIP events), on the decoder shapes of DREAM-resnet (b = 32, 128) and the
post-upsample convs of DREAM-vgg-Q (b = 128).  TFLOP/s are DIRECT sub-pixel FLOPs (16 taps per input pixel and channel pair).
Usage: python tools/microbench_convT4.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import ops  # noqa: E402


def main():
    for (b, h, c) in [(32, 200, 256), (32, 100, 256), (32, 50, 256), (32, 25, 256), (32, 13, 256), (128, 50, 256), (128, 100, 256), (16, 100, 256)]:
        x = torch.randn(b, h, h, c, device="cuda").relu_()
        wT = torch.randn(c, c, 4, 4, device="cuda") * 0.03
        bias = torch.randn(c, device="cuda") * 0.05
        u2, _ = ops.pack_convT4x4_winograd_weight(wT)
        u4, _ = ops.pack_convT4x4_winograd4_weight(wT)
        fns = {"F(2x2)": lambda: ops.conv_transpose4x4s2_winograd(x, u2, c, None, bias, ops.CONV_RELU),
               "F(4x4)": lambda: ops.conv_transpose4x4s2_winograd4(x, u4, c, None, bias, ops.CONV_RELU)}
        # schedule variants of the F(4x4) kernel: DREAM_ALT_LIBS="name=path.so,..." (libraries built with other -DDREAM_W4_PAT_* values)
        y_alt = torch.empty(b, 2 * h, 2 * h, c, device="cuda")
        for item in filter(None, os.environ.get("DREAM_ALT_LIBS", "").split(",")):
            name, path = item.split("=")
            import ctypes
            from dream_amd import _hip
            fn = ctypes.CDLL(path).dream_conv_transpose4x4s2_winograd4_nhwc_f32
            fn.restype, fn.argtypes = _hip._SIGNATURES["dream_conv_transpose4x4s2_winograd4_nhwc_f32"]
            fns[name] = lambda fn=fn: fn(x.data_ptr(), u4.data_ptr(), None, bias.data_ptr(), y_alt.data_ptr(), b, h, h, c, c, 1, torch.cuda.current_stream().cuda_stream)
        y2, y4 = fns["F(2x2)"](), fns["F(4x4)"]()
        torch.cuda.synchronize()
        diff = float((y2 - y4).abs().max() / y2.abs().max())
        best = {k: 1e9 for k in fns}
        for _ in range(5):
            for k, fn in fns.items():
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn()
                e.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], s.elapsed_time(e))
        fl = 2.0 * 16 * c * c * h * h * b
        for k in fns:
            if k not in ("F(2x2)", "F(4x4)"):
                print("      %-24s %7.3f ms (%.2f of peak)" % (k, best[k], fl * (25.0 / 64.0) / best[k] / 1e9 / 157.3))
        print("b=%3d %3dx%-3d %d->%d  F(2x2) %7.3f ms %6.1f TF | F(4x4) %7.3f ms %6.1f TF (%.2f of peak on its own MACs)  speedup %.2f  max diff / max %.1e"
              % (b, h, h, c, c, best["F(2x2)"], fl / best["F(2x2)"] / 1e9, best["F(4x4)"], fl / best["F(4x4)"] / 1e9,
                 fl * (25.0 / 64.0) / best["F(4x4)"] / 1e9 / 157.3, best["F(2x2)"] / best["F(4x4)"], diff), flush=True)
        # the data gradient (dy [b,2h,2h,c] -> dx [b,h,h,c]): four phase convs on stride-2 views of dy, summed
        dy = torch.randn(b, 2 * h, 2 * h, c, device="cuda")
        ub2, _ = ops.pack_convT4x4_winograd_weight(wT, 1)
        ub4, _ = ops.pack_convT4x4_winograd4_weight(wT, 1)
        gf = {"F(2x2)": lambda: ops.conv4x4s2_winograd(dy, ub2, c), "F(4x4)": lambda: ops.conv4x4s2_winograd4(dy, ub4, c)}
        d2, d4 = gf["F(2x2)"](), gf["F(4x4)"]()
        torch.cuda.synchronize()
        gdiff = float((d2 - d4).abs().max() / d2.abs().max())
        gbest = {k: 1e9 for k in gf}
        for _ in range(5):
            for k, fn in gf.items():
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn()
                e.record()
                torch.cuda.synchronize()
                gbest[k] = min(gbest[k], s.elapsed_time(e))
        print("      data gradient:     F(2x2) %7.3f ms %6.1f TF | F(4x4) %7.3f ms %6.1f TF  speedup %.2f  max diff / max %.1e"
              % (gbest["F(2x2)"], fl / gbest["F(2x2)"] / 1e9, gbest["F(4x4)"], fl / gbest["F(4x4)"] / 1e9, gbest["F(2x2)"] / gbest["F(4x4)"], gdiff), flush=True)
        del x, y2, y4, dy, d2, d4


if __name__ == "__main__":
    main()
