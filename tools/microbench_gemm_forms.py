#!/usr/bin/env python
"""The 1x1-conv GEMM of ResNet-101's Bottlenecks in the FORMS the training step launches (plain, batch statistics in the epilogue,
BatchNorm + ReLU in the loader, the data gradient with mask / residual / reductions), on WARM operands (one buffer set, re-used: it
sits in the 256-MB Infinity Cache -- what tools/microbench_conv1x1.py measures) and on COLD ones (a rotation of buffer sets larger
than the cache: what the step sees).  Round 6: the in-step launches ran at 0.24-0.51 of the MFMA peak while the warm micro-benchmark
reported 0.55-0.65.

    python tools/microbench_gemm_forms.py [--batch 16] [--reps 5] [--shapes layer3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dream_amd import ops  # noqa: E402

SHAPES = {
    "layer1": [(100, 64, 256), (100, 256, 64)],
    "layer2": [(50, 128, 512), (50, 512, 128)],
    "layer3": [(25, 256, 1024), (25, 1024, 256)],
    "layer4": [(13, 512, 2048), (13, 2048, 512)],
}


def time_rotation(fns, reps):
    """fns: one closure per buffer set.  Returns microseconds per launch (best of `reps` passes over all sets)."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        n = 0
        for _ in range(max(1, 24 // len(fns))):
            for f in fns:
                f()
                n += 1
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shapes", default="layer3,layer2,layer4,layer1")
    ap.add_argument("--cold-mb", type=int, default=768)
    a = ap.parse_args()
    dev = torch.device("cuda")
    ctr = ops.bn_counter_buffer(dev)
    print("form                      res   K ->    N |  warm us   TF  frac |  cold us   TF  frac | cold MB/launch  GB/s", flush=True)
    for name in a.shapes.split(","):
        for (res, cin, cout) in SHAPES[name]:
            m = a.batch * res * res
            flops = 2.0 * m * cin * cout
            w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
            pf, _ = ops.pack_conv1x1_weight(w, 0)
            pt, _ = ops.pack_conv1x1_weight(w, 1)          # data-gradient operator: K = cout, N = cin
            bn = torch.nn.BatchNorm2d(cout).to(dev)
            per_set_mb = m * (cin + 4 * cout) * 4 / 1e6
            nsets = max(2, int(a.cold_mb / per_set_mb) + 1)
            sets = []
            for _ in range(nsets):
                sets.append(dict(x=torch.randn(a.batch, res, res, cin, device=dev), r=torch.randn(a.batch, res, res, cout, device=dev),
                                 dy=torch.randn(a.batch, res, res, cout, device=dev), zin=torch.randn(a.batch, res, res, cin, device=dev),
                                 yact=torch.randn(a.batch, res, res, cin, device=dev), rin=torch.randn(a.batch, res, res, cin, device=dev)))
            pre_ab = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1])
            abz = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1])
            mean, invstd = torch.randn(cin, device=dev) * 0.1, torch.rand(cin, device=dev) + 0.5
            sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
            forms = [
                # (label, bytes per launch, closure factory)
                ("fwd plain+res+relu", m * (cin + 2 * cout) * 4,
                 lambda s: (lambda: ops.conv1x1(s["x"], pf, cout, sc, sh, s["r"], ops.CONV_RELU))),
                ("fwd plain", m * (cin + cout) * 4,
                 lambda s: (lambda: ops.conv1x1(s["x"], pf, cout))),
                ("fwd stats (EPI1)", m * (cin + cout) * 4,
                 lambda s: (lambda: ops.conv1x1_bn(s["x"], pf, cout, bn, ctr))),
                ("fwd pre+stats (PRE,EPI1)", m * (cin + cout) * 4,
                 lambda s: (lambda: ops.conv1x1_bn(s["x"], pf, cout, bn, ctr, pre_ab=pre_ab))),
                ("bwd mask(z,ab) (EPI2)", m * (cout + 2 * cin) * 4,
                 lambda s: (lambda: ops.conv1x1_bwd_bnmask(s["dy"], pt, cin, s["zin"], abz, mean, invstd, ctr))),
                ("bwd mask(yact)+res (EPI2)", m * (cout + 4 * cin) * 4,
                 lambda s: (lambda: ops.conv1x1_bwd_bnmask(s["dy"], pt, cin, s["zin"], None, mean, invstd, ctr, y_act=s["yact"],
                                                          residual=s["rin"]))),
            ]
            def keep(s, fn):
                # the result stays referenced by its buffer set until that set's next launch: outputs rotate like the inputs
                def run():
                    s["out"] = None
                    s["out"] = fn()
                return run

            for label, nbytes, make in forms:
                warm = time_rotation([keep(sets[0], make(sets[0]))], a.reps)
                cold = time_rotation([keep(s, make(s)) for s in sets], a.reps)
                for s in sets:
                    s["out"] = None
                # (the backward forms contract over cout and produce cin: same FLOPs)
                print("%-26s %3d %4d -> %4d | %7.1f %5.1f %5.2f | %7.1f %5.1f %5.2f | %8.1f %8.0f" % (
                    label, res, cin, cout, warm, flops / warm / 1e6, flops / warm / 1e6 / 157.3, cold, flops / cold / 1e6,
                    flops / cold / 1e6 / 157.3, nbytes / 1e6, nbytes / cold / 1e3), flush=True)
            del sets
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
