#!/usr/bin/env python
"""sha256 of what one BASELINE configuration computes on seeded inputs -- the belief maps of an inference pass, or the loss and the updated
parameters after two training steps -- to compare two library builds bit for bit (tools/gpu_round.sh stages copy build/lib<X>.so over the
product library in turn).      python tools/digest_step.py --arch resnet_h --mode train --batch 16"""
import argparse
import contextlib
import hashlib
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
    ap.add_argument("--mode", default="infer")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--res", type=int, default=400)
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
    h = hashlib.sha256()
    if a.mode == "train":
        net.enable_training()
        ow, oh = net.trained_net_output_resolution()
        tgt = torch.from_numpy(cases.target_batch(a.batch, n_kp, (ow, oh), in_wh=(a.res, a.res), seed=0)).cuda()
        for _ in range(2):
            loss = net.train([x], tgt)
            h.update(torch.as_tensor(loss).detach().cpu().numpy().tobytes())
        for k, v in sorted(net.model.state_dict().items()):
            h.update(v.detach().cpu().numpy().tobytes())
    else:
        net.enable_evaluation()
        with torch.no_grad():
            maps, kps = net.inference(x)
        h.update(maps.cpu().numpy().tobytes())
        h.update(kps.cpu().numpy().tobytes())
    print("%s %s b=%d sha256 %s" % (a.arch, a.mode, a.batch, h.hexdigest()[:24]))


if __name__ == "__main__":
    main()
