#!/usr/bin/env python
"""Do the replayed training steps of the single-process data-parallel path depend on WHEN the GPU runs them?  Six ResNet training steps
with gpu_ids=[0, 0, 0, 0] (eager, capture + replay, four replays) per run, several runs per setting, the losses printed as hex floats:
every run of every setting must print the line of the eager steps.  The settings move a device-wide synchronisation through the step
(DREAM_DP_PROBE_SYNC, data_parallel._probe_sync): it changes nothing but the timing.

Round 6 found two faults with it (profiles/r06_dp_exchange_probe.txt):
  * a hipMemsetAsync captured into a hipGraph (a memset NODE) is not reliably ordered with the kernel nodes around it on this runtime
    (ROCm 7.2; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 hides it): after a device synchronisation between two replays the stride-2 1x1 data
    gradient read its "zeroed" output as the previous owner of the memory had left it -- 1e20-sized gradients from layer4.0 down.
    The library zeroes and copies with kernels now (csrc/common.h: dream_zero_words / dream_copy_words);
  * streams drawn from torch's pool of 32 alias each other in a long-running process -- the sixth network of one process captured
    on an alias of the stream another replica was replaying to.  The package keeps streams of its own (_hip.own_stream).

    python tools/dp_exchange_probe.py [runs] > gpurun_out/r06_probe/dp_exchange_probe.txt
"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch

import dream_amd
from golden import cases
import oracle.models as om


def network(wts, replicas):
    cfg = dream_amd.default_network_config("resnet_h", "panda", optimizer="adam", learning_rate=1e-5)
    cfg["training"]["config"]["net_input_resolution"] = [64, 64]
    cfg["training"]["platform"]["gpu_ids"] = [0] * replicas
    with contextlib.redirect_stdout(io.StringIO()):
        net = dream_amd.create_network_from_config_data(cfg)
    net.model.load_state_dict({"module." + k: v for k, v in wts.items()})
    return net


def run(wts, env, steps=6, replicas=4, frames=8):
    """-> (losses of ``steps`` training steps under ``env``, the model's counters)."""
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        net = network(wts, replicas)
        net.enable_training()
        ow, oh = net.trained_net_output_resolution()
        x = torch.from_numpy(cases.image_batch(frames, 64, 64, seed=41)).to("cuda:0")
        t = torch.from_numpy(cases.target_batch(frames, 7, (ow, oh), in_wh=(64, 64), seed=41)).to("cuda:0")
        losses = [net.train([x], t).item() for _ in range(steps)]
        torch.cuda.synchronize()
        return losses, dict(net.model.stats)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


GRAPHS = {"DREAM_DP_GRAPHS": "1"}
SETTINGS = [
    ("eager, exchange in one piece", {"DREAM_DP_GRAPHS": "0", "DREAM_DP_BUCKETS": "0"}),
    ("eager, two pieces", {"DREAM_DP_GRAPHS": "0"}),
    ("graphs, one piece", dict(GRAPHS, DREAM_DP_BUCKETS="0")),
    ("graphs, two pieces", GRAPHS),
    ("graphs, one piece, sync before the backward", dict(GRAPHS, DREAM_DP_BUCKETS="0", DREAM_DP_PROBE_SYNC="backward_begin")),
    ("graphs, one piece, sync after the backward", dict(GRAPHS, DREAM_DP_BUCKETS="0", DREAM_DP_PROBE_SYNC="backward_end")),
    ("graphs, two pieces, sync before the exchange", dict(GRAPHS, DREAM_DP_PROBE_SYNC="exchange_begin")),
    ("graphs, two pieces, sync between the pieces", dict(GRAPHS, DREAM_DP_PROBE_SYNC="exchange_mid")),
    ("graphs, two pieces, sync after the exchange", dict(GRAPHS, DREAM_DP_PROBE_SYNC="exchange_end")),
    ("graphs (8 leaves a segment), sync after the backward", dict(GRAPHS, DREAM_TRAIN_GRAPH_SPLIT="8", DREAM_DP_PROBE_SYNC="backward_end")),
    ("one backward graph, sync after the backward", dict(GRAPHS, DREAM_TRAIN_GRAPH_SPLIT="0", DREAM_DP_PROBE_SYNC="backward_end")),
]


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
    wts = om.recipe_weights(om.build_model("resnet_h", 7).state_dict(), ("upsample.12.weight", "upsample.12.bias"), 0.1)
    ref, bad = None, 0
    for name, env in SETTINGS:
        for r in range(runs):
            losses, stats = run(wts, env)
            line = " ".join(float(v).hex() for v in losses)
            ref = ref or line
            bad += line != ref
            print("%-54s run %d  %s  %s  exchanges %s replays %s" % (name, r, "same" if line == ref else "DIFFERENT", line,
                                                                      stats.get("exchanges"), stats.get("replays")), flush=True)
    print("%d of %d runs differ from the eager steps" % (bad, runs * len(SETTINGS)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
