#!/usr/bin/env python
"""Knock-out timing of the LDS-staged Winograd weight-gradient kernel (bit 0: no global loads, bit 1: no transform pieces,
bit 2: no per-stage barrier -- results wrong by construction; bit 3: no stagger of the transform pieces between the two waves of a SIMD).
python tools/wgw_diag.py build | run"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KS = [1, 2, 3, 4]
OUT = os.path.join(ROOT, "build", "diag")


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, "dream_amd", "csrc")
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                               "-shared", "-I", os.path.join(csrc, "include"), "-DDREAM_WGW_DIAG=%d" % k,
                               os.path.join(csrc, "wgrad_wino.hip"), os.path.join(csrc, "api.hip"), "-o",
                               os.path.join(OUT, "libwgw_diag_%d.so" % k)]) for k in KS]
    assert all(p.wait() == 0 for p in procs)


def run():
    import torch
    from dream_amd import _hip
    libs = {0: _hip.lib()}
    for k in KS:
        libs[k] = ctypes.CDLL(os.path.join(OUT, "libwgw_diag_%d.so" % k))
    names = {0: "product", 1: "no loads", 2: "no transforms", 3: "MFMAs + operand reads only", 4: "no barrier", 8: "producer work of both halves at the same MFMA numbers"}
    for (b, res, cin, cout) in [(128, 400, 64, 64), (128, 100, 256, 256), (128, 50, 512, 512)]:
        x = torch.randn(b, res, res, cin, device="cuda")
        dy = torch.randn(b, res, res, cout, device="cuda")
        dw = torch.empty(cout, cin, 3, 3, device="cuda")
        nbytes = int(libs[0].dream_conv3x3_wgrad_winograd_workspace(b, res, res, cin, cout))
        ws = torch.empty(nbytes // 4, device="cuda")
        flops = 2.0 * b * res * res * cin * cout * 9 / 2.25
        line = []
        for k in [0] + KS:
            fn = libs[k].dream_conv3x3_wgrad_winograd_nhwc_f32
            fn.restype, fn.argtypes = _hip._SIGNATURES["dream_conv3x3_wgrad_winograd_nhwc_f32"]

            def call():
                assert fn(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), b, res, res, cin, cout, cout, 0,
                          torch.cuda.current_stream().cuda_stream) == 0
            call()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                call()
                e.record()
                torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e))
            line.append("%s %.3f ms (%.2f)" % (names[k], best, flops / best / 1e9 / 157.3))
        print("%d %d->%d: " % (res, cin, cout) + " | ".join(line), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
