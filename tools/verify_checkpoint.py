#!/usr/bin/env python
"""Check a released / trained DREAM checkpoint against this implementation and run it once.

    python tools/verify_checkpoint.py <network.yaml> <network.pth> [--image frame.png] [--device cuda:0]

What the reference does with such a pair: ``create_network_from_config_file(yaml, pth)`` (dream/network.py:29-63), i.e.
``DreamNetwork(config)`` then ``model.load_state_dict(torch.load(pth))`` (also dream/analysis.py:148,
scripts/network_inference.py:97).  This tool
  1. builds the network from the YAML on the HIP path and diffs the checkpoint's manifest (keys, shapes, dtypes) against the
     model's ``state_dict()``: missing keys, unexpected keys, shape and dtype mismatches are listed and fail the run;
  2. loads the checkpoint strictly and reads it back (every tensor bit-identical after the round trip through the device);
  3. with a GPU: one ``inference`` on a synthetic frame (or ``--image``, through ``keypoints_from_image``), checks that the
     belief maps are finite and have the trained output resolution, and prints the keypoints; a second call must reproduce
     them bit for bit.
Exit status 0 = the checkpoint is usable as is.  Nothing here needs the reference or torchvision."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dream_amd  # noqa: E402
from dream_amd import network as dnet  # noqa: E402


def manifest_diff(model_sd, ckpt_sd):
    """-> list of human-readable problems (empty = identical manifests)."""
    problems = []
    for k in model_sd:
        if k not in ckpt_sd:
            problems.append("missing in checkpoint: %s %s" % (k, tuple(model_sd[k].shape)))
    for k, v in ckpt_sd.items():
        if k not in model_sd:
            problems.append("unexpected in checkpoint: %s %s" % (k, tuple(v.shape)))
        elif tuple(v.shape) != tuple(model_sd[k].shape):
            problems.append("shape mismatch: %s checkpoint %s, model %s" % (k, tuple(v.shape), tuple(model_sd[k].shape)))
        elif v.dtype != model_sd[k].dtype:
            problems.append("dtype mismatch: %s checkpoint %s, model %s" % (k, v.dtype, model_sd[k].dtype))
    return problems


def verify(config_path, params_path, image_path=None, device=None, out=print):
    cfg = dnet._load_yaml(config_path)
    if device is not None and device.startswith("cuda"):
        idx = int(device.split(":")[1]) if ":" in device else 0
        cfg.setdefault("training", {}).setdefault("platform", {})["gpu_ids"] = [idx]
    net = dream_amd.create_network_from_config_data(cfg)
    ckpt = torch.load(params_path, map_location="cpu")
    if not isinstance(ckpt, dict):
        out("FAIL: %s does not hold a state_dict (got %s)" % (params_path, type(ckpt).__name__))
        return 1
    model_sd = net.model.state_dict()
    problems = manifest_diff(model_sd, ckpt)
    n_param = sum(v.numel() for k, v in ckpt.items() if v.dtype.is_floating_point and "running_" not in k)
    out("checkpoint: %d tensors, %d parameters; model (%s): %d tensors" % (
        len(ckpt), n_param, type(net.model.module).__name__, len(model_sd)))
    if problems:
        for p in problems[:40]:
            out("  " + p)
        out("FAIL: %d manifest problem(s)" % len(problems))
        return 1
    out("manifest: keys, shapes and dtypes agree")
    net.model.load_state_dict(ckpt, strict=True)
    back = net.model.state_dict()
    bad = [k for k, v in ckpt.items() if not torch.equal(back[k].cpu(), v)]
    if bad:
        out("FAIL: %d tensors changed by load_state_dict/state_dict, e.g. %s" % (len(bad), bad[0]))
        return 1
    out("round trip: every tensor bit-identical after load_state_dict -> state_dict")
    if not torch.cuda.is_available():
        out("no GPU visible: inference check skipped (the HIP path has no CPU fallback)")
        return 0
    net.enable_evaluation()
    w, h = net.trained_net_input_resolution()
    ow, oh = net.trained_net_output_resolution()
    with torch.no_grad():
        if image_path:
            from PIL import Image
            res = net.keypoints_from_image(Image.open(image_path).convert("RGB"), debug=True)
            maps, kps = res["belief_maps"][None], res["detected_keypoints"]
            again = net.keypoints_from_image(Image.open(image_path).convert("RGB"))["detected_keypoints"]
        else:
            rs = np.random.RandomState(0)
            u8 = rs.randint(0, 256, (1, h, w, 3)).astype(np.uint8)
            x = torch.from_numpy(((u8.astype(np.float32) / 255.0 - 0.5) / 0.5).transpose(0, 3, 1, 2).copy())
            maps, kps = net.inference(x)
            again = net.inference(x)[1].numpy()
            kps = kps.numpy()
    if tuple(maps.shape[-2:]) != (oh, ow) and not image_path:
        out("FAIL: belief maps are %s, trained output resolution is %s" % (tuple(maps.shape[-2:]), (oh, ow)))
        return 1
    if not bool(torch.isfinite(maps).all()):
        out("FAIL: non-finite belief map values")
        return 1
    if not np.array_equal(np.asarray(kps), np.asarray(again)):
        out("FAIL: two inference calls disagree")
        return 1
    out("inference: belief maps %s finite, max %.4f; keypoints (%s):" % (
        tuple(maps.shape), float(maps.max()), "raw image frame" if image_path else "net output frame"))
    for name, kp in zip(net.keypoint_names, np.asarray(kps).reshape(-1, 2)):
        out("  %-24s %10.3f %10.3f" % (name, kp[0], kp[1]))
    out("OK")
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("config")
    ap.add_argument("params")
    ap.add_argument("--image", default=None, help="RGB frame to run through keypoints_from_image")
    ap.add_argument("--device", default=None, help="e.g. cuda:0 (default: the YAML's gpu_ids / current device)")
    a = ap.parse_args(argv)
    return verify(a.config, a.params, a.image, a.device)


if __name__ == "__main__":
    sys.exit(main())
