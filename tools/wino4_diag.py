#!/usr/bin/env python
"""Knock-out timing of the Winograd F(4x4,3x3) kernel: libraries built with -DDREAM_W4_DIAG=k (bit 0: no patch loads, bit 1: no
weight stream, bit 2: no barriers, bit 3: no pass 1 / pass 2, bit 7: no epilogue, bit 8: epilogue without stores; results are wrong by construction) against the product library, same
layer, same box.  `build` runs here (hipcc), `run` on the GPU box.   python tools/wino4_diag.py build | run [--batch 128]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KS = [128, 256, 15, 2, 32, 64, 96]          # 1000 + r: product code with weight-ring size r; 2001..: schedule variants
if os.environ.get("DREAM_W4_DIAG_KS"):          # a subset of the variants: DREAM_W4_DIAG_KS=1008,128
    KS = [int(v) for v in os.environ["DREAM_W4_DIAG_KS"].split(",")]
OUT = os.path.join(ROOT, "build", "diag")             # travels with the snapshot only while it exists: `rm -rf build/diag` after the measurement
SCALAR = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DDREAM_PACKED_F32=0"]     # __graft_entry__.SCALAR_F32_FLAGS
VARIANTS = {6002: SCALAR + ["-DDREAM_W4_WAUX=2"], 6001: SCALAR + ["-DDREAM_W4_WAUX=1"], 6016: SCALAR + ["-DDREAM_W4_WAUX=16"], 6102: SCALAR + ["-DDREAM_W4_XAUX=2"],
            6116: SCALAR + ["-DDREAM_W4_XAUX=16"], 6202: SCALAR + ["-DDREAM_W4_WAUX=2", "-DDREAM_W4_XAUX=2"], 5003: SCALAR + ["-DDREAM_W4_STAG_LX=3"], 5004: SCALAR + ["-DDREAM_W4_STAG_LX=4"], 5005: SCALAR + ["-DDREAM_W4_STAG_LX=5"], 5006: SCALAR + ["-DDREAM_W4_STAG_LX=6"],
            5007: SCALAR + ["-DDREAM_W4_STAG_LX=7"], 4001: SCALAR, 4002: SCALAR + ["-DDREAM_W4_SETPRIO=1"], 4003: ["-DDREAM_PACKED_F32=1", "-DDREAM_W4_RUNNING_WOFF=0"], 4004: ["-DDREAM_PACKED_F32=1", "-DDREAM_W4_SETPRIO=1"],
            4005: ["-DDREAM_PACKED_F32=1"], 3001: ["-DDREAM_W4_STORE=buffer_store_f32_nt"], 2001: ["-DDREAM_W4_S1=6", "-DDREAM_W4_S2=9", "-DDREAM_W4_LX=3"], 2002: ["-DDREAM_W4_S1=8", "-DDREAM_W4_S2=13", "-DDREAM_W4_LX=3"],
            2003: ["-DDREAM_W4_S1=10", "-DDREAM_W4_S2=13", "-DDREAM_W4_LX=1"], 2004: ["-DDREAM_W4_S1=11", "-DDREAM_W4_S2=14", "-DDREAM_W4_LX=3"],
            2005: ["-DDREAM_W4_S1=4", "-DDREAM_W4_S2=8", "-DDREAM_W4_LX=1"], 2101: ["-DDREAM_W4_MIDBARRIER=1"], 2100: ["-DDREAM_W4_MIDBARRIER=0"]}
NAMES = {0: "product", 6002: "weights nt", 6001: "weights sc0", 6016: "weights sc1", 6102: "patches nt", 6116: "patches sc1", 6202: "weights + patches nt", 5003: "wavefronts 4-7: patch loads 3 slots later", 5004: "... 4 slots later", 5005: "... 5 slots later", 5006: "... 6 slots later", 5007: "... 7 slots later", 4001: "scalar fp32 VALU (no v_pk_*)", 4002: "scalar fp32 VALU + s_setprio 1 on waves 4-7", 4003: "round-4 code (packed, table weight offsets)",
         4004: "packed + s_setprio 1 on waves 4-7", 4005: "packed fp32 VALU", 3001: "non-temporal stores", 128: "no epilogue", 256: "epilogue without stores", 143: "MFMAs + operand reads only, no epilogue", 32: "weights from L1 (one position)", 64: "patches: chunk 0 only", 96: "weights from L1 + patches chunk 0", 48: "weights from L1 + patches out of range", 2001: "S1 6 S2 9", 2002: "S1 8 S2 13", 2003: "S1 10 S2 13, loads in slot 0", 2004: "S1 11 S2 14",
         2005: "S1 4 S2 8, loads in slot 0", 2101: "with the mid-chunk workgroup barrier (round 3)", 2100: "without the mid-chunk barrier", 16: "patch loads out of range", 18: "patch loads out of range, no weight stream", 1: "no patch loads", 2: "no weight stream", 4: "no barriers", 8: "no passes (loads kept)", 9: "no loads, no passes",
         11: "no loads / passes / weights", 15: "MFMAs + operand reads only", 1006: "weight ring 6 (4 ahead; the product has 8)", 1112: "narrow shape: weight ring 12 (product 18)"}


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, "dream_amd", "csrc")
    procs = []
    for k in KS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-I", os.path.join(csrc, "include"), "-DDREAM_W4_DIAG=%d" % (k if k < 1000 else 0)] + (
                   ["-DDREAM_W4_RING=%d" % (k - 1000)] if 1000 <= k < 1100 else ["-DDREAM_W4_NARROW_RING=%d" % (k - 1100)] if 1100 <= k < 2000 else []) + VARIANTS.get(k, []) + [os.path.join(csrc, "conv_wino4.hip"),
               os.path.join(csrc, "conv_wino.hip"), os.path.join(csrc, "api.hip"), "-o", os.path.join(OUT, "libwino4_diag_%d.so" % k)]
        procs.append(subprocess.Popen(cmd))
    assert all(p.wait() == 0 for p in procs)


def run(batch):
    import torch
    from dream_amd import _hip, ops
    libs = {0: _hip.lib()}
    for k in KS:
        h = ctypes.CDLL(os.path.join(OUT, "libwino4_diag_%d.so" % k))
        fn = h.dream_conv3x3_winograd4_nhwc_f32
        fn.restype, fn.argtypes = _hip._SIGNATURES["dream_conv3x3_winograd4_nhwc_f32"]
        libs[k] = h
    for (res, cin, cout) in [(400, 64, 64), (200, 64, 128), (200, 128, 128), (100, 256, 256), (50, 512, 512), (25, 512, 512)]:
        x = torch.randn(batch, res, res, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        u, _ = ops.pack_weight_winograd4(w, 0)
        y = torch.empty(batch, res, res, cout, device="cuda")
        flops = 2.0 * batch * res * res * cin * cout * 9 / 4.0
        # round-robin over the variants, minimum per variant: whatever runs first after a pause is a few per cent slower (clocks),
        # which a variant-by-variant loop books to the first variant
        calls = {}
        for k in [0] + KS:
            fn = libs[k].dream_conv3x3_winograd4_nhwc_f32

            def call(fn=fn):
                rc = fn(x.data_ptr(), u.data_ptr(), None, None, None, y.data_ptr(), batch, res, res, cin, cout, 1,
                        torch.cuda.current_stream().cuda_stream)
                assert rc == 0
            call()
            calls[k] = call
        torch.cuda.synchronize()
        best = {k: 1e9 for k in calls}
        for _ in range(6):
            for k, call in calls.items():
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                call()
                e.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], s.elapsed_time(e))
        line = ["%s %.3f ms (%.2f)" % (NAMES[k], best[k], flops / best[k] / 1e9 / 157.3) for k in calls]
        print("%d %d->%d b=%d: " % (res, cin, cout, batch) + " | ".join(line), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 128)
