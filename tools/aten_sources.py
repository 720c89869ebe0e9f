#!/usr/bin/env python
"""Which lines of dream_amd still reach ATen / runtime data movement inside a step?  Runs a few steps of one BASELINE configuration under
torch.profiler (with_stack) and prints, for every aten:: operator that launches a device kernel or copy (copy_, clone, fill_, zero_, cat,
add_, mul ...), calls per step and the innermost dream_amd frame that issued it.

    python tools/aten_sources.py --arch resnet_h --mode train --batch 16
"""
import argparse
import collections
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet_h")
    ap.add_argument("--mode", default="train")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    sys.argv = sys.argv[:1]
    import bench
    import cases
    import dream_amd
    n_kp, manip = bench.ARCH_K[a.arch]
    cfg = dream_amd.default_network_config(a.arch, manip, batch_size=a.batch)
    cfg["training"]["config"]["net_input_resolution"] = [a.res, a.res]
    with contextlib.redirect_stdout(io.StringIO()):
        net = dream_amd.create_network_from_config_data(cfg)
    net.model.load_state_dict(bench.synthetic_weights(net.model.state_dict()))
    x = torch.from_numpy(cases.image_batch(a.batch, a.res, a.res, seed=0)).cuda()
    if a.mode == "train":
        net.enable_training()
        ow, oh = net.trained_net_output_resolution()
        tgt = torch.from_numpy(cases.target_batch(a.batch, n_kp, (ow, oh), in_wh=(a.res, a.res), seed=0)).cuda()
    else:
        net.enable_evaluation()
        net.hip_graph = False

    def step():
        if a.mode == "train":
            return net.train([x], tgt)
        with torch.no_grad():
            return net.inference(x)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    table = collections.Counter()
    dev_us = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
            continue                                        # outermost ATen call only
        dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        if not dt:
            continue                                        # views, reshapes: nothing on the device
        where = "(no dream_amd frame)"
        for fr in ev.stack or []:
            if "dream_amd" in fr or "bench.py" in fr:
                where = fr.strip().replace(os.path.abspath(ROOT) + "/", "")
                break
        table[(ev.name, where)] += 1
        dev_us[(ev.name, where)] += dt
    print("%s %s b=%d: ATen operators with device work, per step (%d steps traced)" % (a.arch, a.mode, a.batch, a.steps))
    for (name, where), n in sorted(table.items(), key=lambda kv: -dev_us[kv[0]]):
        print("%-22s %7.1f calls/step %9.1f us/step  %s" % (name, n / a.steps, dev_us[(name, where)] / a.steps, where[:150]))
    print("total: %.1f calls/step, %.1f us of device time per step" % (sum(table.values()) / a.steps, sum(dev_us.values()) / a.steps))


if __name__ == "__main__":
    main()
