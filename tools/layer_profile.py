#!/usr/bin/env python
"""Where does a step go, operator by operator?  Wraps every function of dream_amd.ops with HIP-event timing (each call is
synchronised, so the numbers are kernel time without launch overlap), runs a few steps of one BASELINE configuration and prints
one line per (operator, tensor shapes): calls/step, ms/step and -- for the contractions -- TFLOP/s of direct-convolution FLOPs.

    python tools/layer_profile.py --arch resnet_h --mode train --batch 16 [--steps 2]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def flops_of(name, args, out):
    """Direct-convolution FLOPs of one call, or 0 for the streaming operators."""
    t = [a for a in args if torch.is_tensor(a)]
    ints = [a for a in args if isinstance(a, int) and not isinstance(a, bool)]
    try:
        if name in ("conv3x3", "conv3x3_winograd"):
            x = t[0]
            cout = ints[0]
            return 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] * cout * 9
        if name in ("conv1x1", "conv1x1_bn", "conv1x1_bwd_bnmask"):
            x = t[0]
            return 2.0 * x.numel() * ints[0]
        if name == "conv1x1_wgrad":
            x = t[0]
            return 2.0 * x.numel() * ints[0]
        if name == "conv2d" or name == "conv2d_amax":
            x = t[0]
            cout, k = ints[0], ints[1]
            o = out[0] if isinstance(out, tuple) else out
            return 2.0 * o.shape[0] * o.shape[1] * o.shape[2] * cout * x.shape[3] * k * k if o.dim() == 4 and o.shape[3] >= cout \
                else 2.0 * o.numel() * x.shape[3] * k * k
        if name in ("conv_transpose4x4s2", "conv_transpose4x4s2_winograd"):
            x = t[0]
            return 2.0 * x.numel() * ints[0] * 16
        if name == "conv4x4s2":
            o = out
            return 2.0 * o.shape[0] * o.shape[1] * o.shape[2] * ints[0] * t[0].shape[3] * 16
        if name == "conv2d_bwd_data":
            dy = t[0]
            cin, k = ints[0], ints[1]
            return 2.0 * dy.numel() * cin * k * k
        if name in ("conv3x3_wgrad", "conv3x3_wgrad_winograd"):
            x, dy = t[0], t[1]
            return 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * ints[0] * ints[1] * 9
        if name == "conv2d_wgrad":
            x, dy = t[0], t[1]
            return 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * ints[0] * ints[1] * ints[2] * ints[2]
        if name in ("convT4x4_wgrad", "convT_wgrad"):
            x, dy = t[0], t[1]
            return 2.0 * x.numel() * dy.shape[3] * 16
    except (IndexError, AttributeError):
        return 0.0
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet_h")
    ap.add_argument("--mode", default="train")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--timing", choices=("queued", "sync"), default="queued",
                    help="queued (default): the step runs IN ORDER on one stream (DREAM_OVERLAP_WGRAD=0) with an event pair around every operator "
                         "call and ONE synchronisation per step -- the host stays ahead, so a pair brackets the call's kernels (plus the gap to the "
                         "previous kernel) and nothing of the launch path; sync: a synchronisation after every call (rounds 2-6: every call then "
                         "carries its own launch latency, 10-25 us -- a third of a 60-us GEMM)")
    a = ap.parse_args()
    if a.timing == "queued":
        os.environ["DREAM_OVERLAP_WGRAD"] = "0"

    sys.argv = sys.argv[:1]
    import bench
    from dream_amd import ops

    table = collections.OrderedDict()
    state = {"on": False, "depth": 0}

    def wrap(name, fn):
        def timed(*args, **kw):
            if not state["on"] or state["depth"]:          # an operator that forwards to another one is timed once, outside
                return fn(*args, **kw)
            state["depth"] = 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                out = fn(*args, **kw)
            finally:
                state["depth"] = 0
            e1.record()
            shapes = tuple(tuple(x.shape) for x in args if torch.is_tensor(x))[:2]
            ints = tuple(x for x in args if isinstance(x, int) and not isinstance(x, bool))[:4]
            key = (name, shapes, ints)
            rec = table.setdefault(key, [0, 0.0, 0.0])
            rec[0] += 1
            rec[2] += flops_of(name, args, out)
            if a.timing == "sync":
                e1.synchronize()
                rec[1] += e0.elapsed_time(e1)
            else:
                queued.append((rec, e0, e1))
            return out
        return timed

    queued = []

    empty = [0.0]                    # what an event pair with NOTHING between costs on the stream (measured below; subtracted per call)

    def resolve():
        torch.cuda.synchronize()
        for rec, e0, e1 in queued:
            rec[1] += max(0.0, e0.elapsed_time(e1) - empty[0])
        del queued[:]

    def calibrate():
        torch.zeros(1 << 20, device="cuda").add_(1.0)                 # something in the queue ahead of the pairs, as in the step
        pairs = []
        for _ in range(400):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            a1.record()
            pairs.append((a0, a1))
        torch.cuda.synchronize()
        v = sorted(a0.elapsed_time(a1) for a0, a1 in pairs)
        return v[len(v) // 2]

    skip = {"round_up", "bump_version", "wgrad_winograd_pays", "new_amax", "conv1x1_applies", "conv1x1_wgrad_applies", "bn_counter_buffer"}
    for name in dir(ops):
        fn = getattr(ops, name)
        if callable(fn) and getattr(fn, "__module__", None) == ops.__name__ and not name.startswith("_") and name not in skip:
            setattr(ops, name, wrap(name, fn))

    import io
    import contextlib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
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

    import time
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    # host enqueue time vs wall time of unhooked steps: is the step launch-bound?
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("unhooked: host enqueue %.2f ms/step, wall %.2f ms/step (5 steps)" % ((t1 - t0) / 5e-3, (t2 - t0) / 5e-3))
    if a.timing == "queued":
        empty[0] = calibrate()
        print("event pair with nothing between: %.1f us (median of 400; subtracted from every call)" % (empty[0] * 1e3))
    state["on"] = True
    for _ in range(a.steps):
        step()
        resolve()
    state["on"] = False

    rows = sorted(table.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows) / a.steps
    by_op = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for (name, _, _), v in rows:
        by_op[name][0] += v[0]
        by_op[name][1] += v[1]
        by_op[name][2] += v[2]
    print("%s %s b=%d: %.2f ms/step inside dream_amd.ops (%s)" % (a.arch, a.mode, a.batch, total,
          "in order on one stream, event pairs, one synchronisation per step" if a.timing == "queued" else "a synchronisation after every call"))
    print("-- by operator")
    for name, v in sorted(by_op.items(), key=lambda kv: -kv[1][1]):
        tf = v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 and v[2] > 0 else 0.0
        print("%-28s %5.0f calls/step %8.3f ms/step %5.1f%% %s" % (name, v[0] / a.steps, v[1] / a.steps, 100.0 * v[1] / a.steps / total,
                                                                   ("%6.1f TFLOP/s" % tf) if tf else ""))
    print("-- by operator and shape (top %d)" % a.top)
    for (name, shapes, ints), v in rows[:a.top]:
        tf = v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 and v[2] > 0 else 0.0
        print("%-24s %-44s %-18s x%-3.0f %8.3f ms/step %s" % (name, " ".join("x".join(map(str, s)) for s in shapes), ",".join(map(str, ints)),
                                                              v[0] / a.steps, v[1] / a.steps, ("%6.1f TF" % tf) if tf else ""))


if __name__ == "__main__":
    main()
