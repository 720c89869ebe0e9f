#!/usr/bin/env python
"""Where does the occasional long step of the ResNet-101 training run stall?  N steps of one configuration with host-side timers around
zero_grad / forward + loss / backward / optimizer step (no synchronisation inside a step; one per step at its END so that every step
starts with an empty queue -- a stall then shows as host time, not as back-pressure), the caching allocator's counters and the process'
context switches / page faults per step.  Prints the steps whose host time exceeds 1.6 x the median, with their breakdown.

    python tools/stall_probe.py --arch resnet_h --batch 16 --steps 400 [--no-sync]
"""
import argparse
import contextlib
import io
import os
import resource
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet_h")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--no-sync", action="store_true", help="do not synchronise at the end of a step (the bench's mode: the host runs ahead)")
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
    net.enable_training()
    ow, oh = net.trained_net_output_resolution()
    tgt = torch.from_numpy(cases.target_batch(a.batch, n_kp, (ow, oh), in_wh=(a.res, a.res), seed=0)).cuda()
    for _ in range(4):
        net.train([x], tgt)
    torch.cuda.synchronize()
    rows = []
    keys = ("num_alloc_retries", "num_device_alloc", "num_device_free", "num_sync_all_streams")
    for i in range(a.steps):
        ms0 = torch.cuda.memory_stats()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        net.optimizer.zero_grad()
        t1 = time.perf_counter()
        loss = net.loss([x], tgt)
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        net.optimizer.step()
        t4 = time.perf_counter()
        if not a.no_sync:
            torch.cuda.synchronize()
        t5 = time.perf_counter()
        ms1 = torch.cuda.memory_stats()
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        rows.append(dict(i=i, host=(t4 - t0) * 1e3, zero=(t1 - t0) * 1e3, fwd=(t2 - t1) * 1e3, bwd=(t3 - t2) * 1e3, opt=(t4 - t3) * 1e3,
                         wait=(t5 - t4) * 1e3, alloc={k: ms1.get(k, 0) - ms0.get(k, 0) for k in keys},
                         nvcsw=ru1.ru_nvcsw - ru0.ru_nvcsw, nivcsw=ru1.ru_nivcsw - ru0.ru_nivcsw, minflt=ru1.ru_minflt - ru0.ru_minflt,
                         utime=(ru1.ru_utime - ru0.ru_utime) * 1e3, stime=(ru1.ru_stime - ru0.ru_stime) * 1e3))
    med = sorted(r["host"] for r in rows)[len(rows) // 2]
    tot = sorted(r["host"] + r["wait"] for r in rows)[len(rows) // 2]
    print("%s train b=%d, %d steps, %s: median host enqueue %.2f ms, median step %.2f ms" % (
        a.arch, a.batch, a.steps, "host runs ahead" if a.no_sync else "synchronised at the end of every step", med, tot))
    m = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
    print("median: zero_grad %.2f  forward+loss %.2f  backward %.2f  optimizer %.2f  | user CPU %.1f ms, system CPU %.1f ms, voluntary switches %d, "
          "involuntary %d, minor faults %d per step" % (m("zero"), m("fwd"), m("bwd"), m("opt"), m("utime"), m("stime"), m("nvcsw"), m("nivcsw"), m("minflt")))
    grew = [r for r in rows if r["alloc"].get("num_device_alloc") or r["alloc"].get("num_device_free") or r["alloc"].get("num_alloc_retries")]
    print("%d steps in which the caching allocator went to the device (hipMalloc / hipFree): %s" % (
        len(grew), [(r["i"], {k: v for k, v in r["alloc"].items() if v}, round(r["host"], 1)) for r in grew][:12]))
    slow = [r for r in rows if r["host"] > 1.6 * med]
    print("%d steps with host time > 1.6 x the median:" % len(slow))
    for r in slow:
        print("  step %3d host %.1f ms = zero_grad %.1f + forward %.1f + backward %.1f + optimizer %.1f (then waited %.1f) | user %.1f ms system %.1f ms "
              "vol.switches %d invol. %d minor faults %d | allocator %s" % (r["i"], r["host"], r["zero"], r["fwd"], r["bwd"], r["opt"], r["wait"], r["utime"],
                                                                            r["stime"], r["nvcsw"], r["nivcsw"], r["minflt"],
                                                                            {k: v for k, v in r["alloc"].items() if v}))


if __name__ == "__main__":
    main()
