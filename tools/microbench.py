#!/usr/bin/env python
"""Per-layer / per-variant timing of the MFMA conv kernel on the DREAM-vgg-Q layer shapes (HIP events).
Writes gpurun_out/microbench.json and prints a table.  Usage: python tools/microbench.py [--batch 32]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import _hip, ops  # noqa: E402

LAYERS = [  # (res, cin, cout, flags, count in vgg_q)
    (400, 64, 64, 1, 1), (200, 64, 128, 1, 1), (200, 128, 128, 1, 1), (100, 128, 256, 1, 1), (100, 256, 256, 1, 3),
    (50, 256, 512, 1, 1), (50, 512, 512, 1, 3), (25, 512, 512, 1, 4), (50, 512, 256, 3, 1), (50, 256, 256, 0, 1),
    (100, 256, 128, 3, 1), (100, 128, 64, 0, 1), (100, 64, 64, 1, 1), (100, 64, 32, 1, 1), (100, 32, 7, 4, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", type=str, default="all")
    args = ap.parse_args()
    lib = _hip.lib()
    nv = lib.dream_conv3x3_num_variants()
    variants = list(range(-1, nv)) if args.variants == "all" else [int(v) for v in args.variants.split(",")]
    results = []
    for (res, cin, cout, flags, count) in LAYERS:
        ups = bool(flags & 2)
        hs = res // 2 if ups else res
        x = torch.randn(args.batch, hs, hs, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        bias = torch.randn(cout, device="cuda")
        packed, rows, _, _ = ops.pack_weight(w, 0)
        flops = 2.0 * args.batch * res * res * cin * cout * 9
        row = {"layer": [res, cin, cout, flags], "count": count, "gflop": flops / 1e9, "variants": {}}
        for v in variants:
            name = lib.dream_conv3x3_variant_name(v).decode()
            lib.dream_conv3x3_set_variant(v)
            try:
                ops.conv3x3(x, packed, bias, cout, flags)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(args.reps):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    ops.conv3x3(x, packed, bias, cout, flags)
                    e.record()
                    torch.cuda.synchronize()
                    best = min(best, s.elapsed_time(e))
                row["variants"][name] = {"ms": best, "tflops": flops / best / 1e9}
            except RuntimeError as err:
                row["variants"][name] = {"error": str(err)[:120]}
            finally:
                lib.dream_conv3x3_set_variant(-1)
        if cin % 32 == 0 and not (flags & 4 and False):
            p16 = ops.pack_conv_weight_f16x3(w, 0)
            amax = ops.absmax(x)
            for v16 in (-1, 0, 1, 2, 4, 7):
                lib.dream_conv_f16x3_set_variant(v16)
                try:
                    ops.conv2d_f16x3(x, amax, p16, cout, 3, None, bias, None, flags)
                    torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(args.reps):
                        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        s.record()
                        ops.conv2d_f16x3(x, amax, p16, cout, 3, None, bias, None, flags)
                        e.record()
                        torch.cuda.synchronize()
                        best = min(best, s.elapsed_time(e))
                    row["variants"]["f16x3_v%d" % v16] = {"ms": best, "tflops": flops / best / 1e9}
                except RuntimeError as err:
                    row["variants"]["f16x3_v%d" % v16] = {"error": str(err)[:120]}
                finally:
                    lib.dream_conv_f16x3_set_variant(-1)
        results.append(row)
        line = "%4d %3d->%3d f%d | " % (res, cin, cout, flags) + " ".join(
            "%s:%5.1f" % (k[-11:], v.get("tflops", -1)) for k, v in row["variants"].items())
        print(line, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w") as f:
        json.dump({"batch": args.batch, "results": results}, f, indent=1)
    # weighted whole-network estimate with the best variant per layer vs the heuristic
    t16 = 0.0
    for r in results:
        ok16 = [v["ms"] for k, v in r["variants"].items() if k.startswith("f16x3") and "ms" in v]
        ok32 = [v["ms"] for k, v in r["variants"].items() if not k.startswith("f16x3") and "ms" in v]
        t16 += (min(ok16) if ok16 else min(ok32)) * r["count"]
    print("best with f16x3: conv time for %d frames = %.1f ms -> %.0f frames/s (convs only)" % (args.batch, t16, args.batch / t16 * 1e3))
    for pick in ("heuristic", "best"):
        t = 0.0
        for r in results:
            ok = {k: v for k, v in r["variants"].items() if "ms" in v and not k.startswith("f16x3")}
            ms = ok["heuristic"]["ms"] if pick == "heuristic" else min(v["ms"] for v in ok.values())
            t += ms * r["count"]
        print("%s: conv time for %d frames = %.1f ms -> %.0f frames/s (convs only)" % (pick, args.batch, t, args.batch / t * 1e3))


if __name__ == "__main__":
    main()

