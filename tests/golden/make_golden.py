"""Generate the committed golden fixtures by running the REAL reference (stub-imported, see
ref_import.py) in the dev container.  Usage:  python tests/golden/make_golden.py

Fixtures are data only (inputs are regenerated from seeds by cases.py; outputs are stored):
  peaks_golden.npz      G1/G2  peaks_from_belief_maps + DreamNetwork.inference selection rule
  softargmax_golden.npz G3     SoftArgmaxPavlo
  cnn_<arch>.npz        G4     model(x)[0] for recipe weights (oracle.models.recipe_weights)
  train_vgg_q.npz       G5     one DreamNetwork.train() step (loss, grad norms, post-Adam samples)
  state_dict_manifest.json G6  key -> shape for the four archs (module.-prefixed, as saved)
  variant_<name>.npz    G7     skip / full_output / soft-argmax / multi-stage hourglasses (--only-variants)
  belief_map_golden.npz G10    create_belief_map (--only-belief-maps)
  peak_rule_golden.npz  G12    selection rule with use_belief_peak_scores / belief_peak_next_best_score changed (--only-peak-rules)
  structured_<arch>.npz G11    blob-like O(1) belief maps end to end: maps, keypoints, calibrated last layer (--only-structured)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cases  # noqa: E402
import ref_import  # noqa: E402
from oracle import models as omodels  # noqa: E402


def flatten_peaks(all_peaks):
    """list[K] of list of (x, y, score, id) -> counts[K], xy float64 [N,2], score float32 [N], id [N]."""
    counts = np.array([len(p) for p in all_peaks], np.int64)
    flat = [q for p in all_peaks for q in p]
    xy = np.array([[q[0], q[1]] for q in flat], np.float64).reshape(-1, 2)
    sc = np.array([q[2] for q in flat], np.float32)
    ids = np.array([q[3] for q in flat], np.int64)
    return counts, xy, sc, ids


def variants(dream):
    """G7: the hourglass constructor branches no shipped YAML reaches (skip connections, full-resolution upsample
    decoder, soft-argmax head, multi-stage) -- inference outputs, every stage's maps, and two training steps."""
    manifest = {}
    for name, (shapes, train) in cases.VARIANT_CASES.items():
        base, over = omodels.VARIANTS[name]
        net = dream.create_network_from_config_data(ref_import.network_config(base, overrides=over))
        sd = net.model.state_dict()
        manifest[name] = {key: list(v.shape) for key, v in sd.items()}
        w = omodels.recipe_weights({key[len("module."):]: v for key, v in sd.items()})
        net.model.load_state_dict({"module." + key: v for key, v in w.items()})
        net.enable_evaluation()
        out = {}
        for (b, h, wd) in shapes:
            x = torch.from_numpy(cases.image_batch(b, h, wd, seed=b * 1000 + h))
            tag = "%dx%dx%d" % (b, h, wd)
            with torch.no_grad():
                res = net.inference(x)
                heads = net.model(x)
            out[tag + "/maps"] = res[0].numpy()
            out[tag + "/keypoints"] = res[1].numpy()
            for i, t in enumerate(heads):
                out[tag + "/head%d" % i] = t.numpy()
            print(name, tag, "->", [tuple(t.shape) for t in heads], "absmax %.3f" % float(res[0].abs().max()))
        if train:
            b, h, wd = cases.VARIANT_TRAIN_SHAPE
            cfg = ref_import.network_config(base, lr=cases.TRAIN_LR["adam"], optimizer="adam", overrides=over)
            cfg["training"]["config"]["net_input_resolution"] = [wd, h]
            net = dream.create_network_from_config_data(cfg)
            w = omodels.recipe_weights({key[len("module."):]: v for key, v in net.model.state_dict().items()},
                                       final_keys=cases.TRAIN_FINAL_KEYS, final_scale=cases.TRAIN_FINAL_SCALE)
            net.model.load_state_dict({"module." + key: v for key, v in w.items()})
            net.enable_training()
            ow, oh = net.trained_net_output_resolution()
            x = torch.from_numpy(cases.image_batch(b, h, wd, seed=7))
            tgt = torch.from_numpy(cases.target_batch(b, 7, (ow, oh), in_wh=(wd, h), seed=7))
            losses = []
            for step in range(2):
                losses.append(net.train([x], tgt).item())
                if step == 0:
                    for key, p in net.model.named_parameters():
                        if p.grad is not None:
                            out["train/gradnorm/" + key] = np.array(float(p.grad.double().norm()))
            out["train/losses"] = np.array(losses, np.float64)
            for key, p in net.model.named_parameters():
                out["train/param_sample/" + key] = p.detach().flatten()[:: max(1, p.numel() // 64)][:64].numpy().copy()
            print(name, "train losses", losses)
        np.savez_compressed(os.path.join(HERE, "variant_%s.npz" % name), **out)
    with open(os.path.join(HERE, "variant_state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=False)


def api_surface(dream):
    """G8: names and parameter lists of the drop-in boundary, read off the reference with inspect --
    data only: what a caller can name, not how it is implemented."""
    import inspect

    def params(fn):
        return [[n, None if q.default is inspect._empty else repr(q.default)]
                for n, q in inspect.signature(fn).parameters.items()]

    def methods(cls):
        return {n: params(f) for n, f in vars(cls).items()          # defined by the class itself, not inherited
                if inspect.isfunction(f) and (not n.startswith("_") or n == "__init__")}

    net = dream.create_network_from_config_data(ref_import.network_config("vgg_q"))
    surface = {
        "dream.network": {
            "functions": {n: params(getattr(dream.network, n)) for n in
                          ("create_network_from_config_file", "create_network_from_config_data")},
            "constants": {"KNOWN_ARCHITECTURES": list(dream.network.KNOWN_ARCHITECTURES),
                          "KNOWN_OPTIMIZERS": list(dream.network.KNOWN_OPTIMIZERS)},
            "DreamNetwork": methods(dream.network.DreamNetwork),
            "DreamNetwork.instance_attributes": sorted(k for k in vars(net) if not k.startswith("_")),
        },
        "dream.models": {c: methods(getattr(dream.models, c)) for c in
                         ("DreamHourglass", "DreamHourglassMultiStage", "ResnetSimple")},
        "dream.spatial_softmax": {"SoftArgmaxPavlo": methods(dream.spatial_softmax.SoftArgmaxPavlo)},
        "dream.image_proc": {n: params(getattr(dream.image_proc, n)) for n in
                             ("peaks_from_belief_maps", "create_belief_map", "resolution_after_preprocessing",
                              "shrink_resolution", "shrink_and_crop_resolution", "preprocess_image",
                              "convert_keypoints_to_raw_from_netin", "convert_keypoints_to_netin_from_netout")
                             if hasattr(dream.image_proc, n)},
    }
    with open(os.path.join(HERE, "api_surface.json"), "w") as f:
        json.dump(surface, f, indent=1, sort_keys=True)
    print("api_surface:", {k: len(v) for k, v in surface.items()})


def keypoint_conversions(dream):
    """G9: the step right after the hot path: net-output -> net-input -> raw-image keypoint
    frames (image_proc.py:135-147,215-260) for every preprocessing type, incl. the -999.999 sentinels."""
    out = {}
    for name, (kps, out_res, in_res, raw_res) in cases.keypoint_conversion_cases().items():
        netin = dream.image_proc.convert_keypoints_to_netin_from_netout(kps.astype(float), out_res, in_res)
        out[name + "/netin"] = np.asarray(netin, np.float64)
        for prep in ("none", "resize", "shrink", "shrink-and-crop"):
            raw = dream.image_proc.convert_keypoints_to_raw_from_netin(netin, in_res, raw_res, prep)
            out[name + "/raw/" + prep] = np.asarray(raw, np.float64)
    np.savez_compressed(os.path.join(HERE, "keypoint_conversion.npz"), **out)
    print("keypoint_conversion:", sorted(out)[:4], "...")


def peak_rules(dream):
    """G12: DreamNetwork.inference with the two public attributes of the selection rule changed (network.py:189-191,
    553-560): use_belief_peak_scores = False, belief_peak_next_best_score in {0.1, 0.3, 0.5}."""
    out = {}
    net_q = dream.create_network_from_config_data(ref_import.network_config("vgg_q"))
    all_cases = cases.peak_cases()
    for name in cases.PEAK_RULE_CASES:
        maps, off = all_cases[name]
        net_q.model = lambda x, _m=maps: [torch.from_numpy(_m)[None]]
        net_q.network_config["training"]["config"]["net_output_resolution"] = [400, 400] if off == 0.0 else [100, 100]
        for tag, (use, thr) in cases.PEAK_RULE_SETTINGS.items():
            net_q.use_belief_peak_scores, net_q.belief_peak_next_best_score = use, thr
            out[name + "/" + tag] = net_q.inference(torch.zeros(1, 3, 8, 8))[1].numpy()
    np.savez_compressed(os.path.join(HERE, "peak_rule_golden.npz"), **out)
    print("peak_rule:", {k: int((v[..., 0] > -999).sum()) for k, v in out.items()})


def belief_maps(dream):
    """G10: create_belief_map (image_proc.py:866-910), the training-target renderer right before the hot path: the
    reference's float64 output for every case of cases.belief_map_cases()."""
    out = {}
    for name, (res, pts, sigma) in cases.belief_map_cases().items():
        ref = dream.image_proc.create_belief_map(res, [tuple(p) for p in pts], sigma=sigma)
        assert ref.dtype == np.float64 and ref.shape == (len(pts), res[1], res[0])
        out[name] = ref
        print("belief_map", name, ref.shape, "drawn", int((ref.reshape(len(pts), -1).max(1) > 0).sum()), "/", len(pts))
    np.savez_compressed(os.path.join(HERE, "belief_map_golden.npz"), **out)


def structured(dream):
    """G11: end-to-end fixture with blob-like belief maps of magnitude O(1), so that the north_star's ABSOLUTE 1e-4 bound
    on the maps, 100 % detection agreement and <= 1e-3 px on the keypoints can be demanded (the recipe-weight fixtures give
    noise-like maps of magnitude ~10 on which the 0.25 rule flips on 1e-5).  All layers but the last carry the weights
    named in cases.STRUCTURED_CASES; the last layer (3x3 conv for vgg_q, 1x1 for resnet_h) gets, per keypoint, a random
    channel mix scaled and shifted so that the map's background sits at 0 and its strongest response at 1 -- calibrated
    here on the reference's own activations and STORED in the fixture.  The fixture is certified decidable: the
    reference's own peak stage gives the same detections, and keypoints within 1e-3 px, on maps perturbed by +-1e-4."""
    from oracle import peaks as opeaks
    only = [a for a in sys.argv[1:] if a in cases.STRUCTURED_CASES]
    for case, (arch, manip, n_kp, last, (b, h, wd), recipe, zero_bg) in cases.STRUCTURED_CASES.items():
        if only and case not in only:
            continue
        cfg = ref_import.network_config(arch, manip)
        cfg["training"]["config"]["net_input_resolution"] = [wd, h]
        net = dream.create_network_from_config_data(cfg)
        sd = {key[len("module."):]: v for key, v in net.model.state_dict().items()}
        w = {"structured": omodels.structured_weights, "smooth": omodels.smooth_weights, "recipe": omodels.recipe_weights}[recipe](sd)
        wk, bk = last + ".weight", last + ".bias"
        k, cin, kh, kw = w[wk].shape
        x_np, centres = cases.structured_input(case)
        x = torch.from_numpy(x_np)
        for mix_seed in range(77, 137):          # the first channel mix that gives both detections and rejections (stored below)
            rs = np.random.RandomState(mix_seed)
            mix = rs.uniform(-0.4 if zero_bg else 0.0, 1.0, (k, cin, 1, 1)) * (rs.uniform(0, 1, (k, cin, 1, 1)) < (0.08 if recipe == "smooth" else 0.35))
            if recipe == "smooth":
                mix[np.arange(k), rs.randint(0, cin, k)] += 0.5         # no empty row
            w[wk] = torch.as_tensor(np.broadcast_to(mix, (k, cin, kh, kw)) / (kh * kw), dtype=torch.float32).contiguous()
            w[bk] = torch.zeros(k)
            net.model.load_state_dict({"module." + key: v for key, v in w.items()})
            net.enable_evaluation()
            with torch.no_grad():
                z = net.model(x)[0].double().numpy()                       # [B,K,Ho,Wo] un-calibrated responses
            bg = np.zeros(k) if zero_bg else np.median(z.transpose(1, 0, 2, 3).reshape(k, -1), axis=1)
            flat = (z - bg[None, :, None, None]).transpose(1, 0, 2, 3).reshape(k, -1)
            ext = flat[np.arange(k), np.abs(flat).argmax(1)]                # strongest deviation, with its sign
            if recipe == "smooth":
                # the bumps (blobs a channel's colour filter likes) are positive and scaled to 1; blobs it dislikes clamp the first
                # layer's ReLU and leave dips below the background, several times deeper for some mixes: the maps span [-6, 1]
                ext = flat.max(1)
            scale = 1.0 / ext
            w[wk] = (w[wk].double() * torch.as_tensor(scale).view(k, 1, 1, 1)).float()
            w[bk] = torch.as_tensor(-bg * scale).float()
            net.model.load_state_dict({"module." + key: v for key, v in w.items()})
            with torch.no_grad():
                maps, kps = net.inference(x)
                net.model.double()
                maps64 = net.model(x.double())[0]
                net.model.float()
            maps, kps = maps.numpy(), kps.numpy()
            det = kps[..., 0] > -999
            print("structured", case, maps.shape, "absmax %.3f" % np.abs(maps).max(), "detections", int(det.sum()), "/", det.size,
                  "fp32-vs-fp64 of the reference itself: %.2e" % float(np.abs(maps - maps64.numpy()).max()))
            if not (cases.STRUCTURED_MIN_DETECTIONS.get(case, 0.25) <= det.mean() <= 0.95):                            # both detections and rejections wanted
                continue
            off0 = opeaks.upsampling_offset(*net.trained_net_output_resolution())
            n_peaks = [[len(pk) for pk in dream.image_proc.peaks_from_belief_maps(torch.from_numpy(maps[i]), off0)] for i in range(b)]
            rejected = int(sum(1 for i in range(b) for kp in range(k) if n_peaks[i][kp] > 1 and not det[i, kp]))
            print("   rejected by the 0.25 rule (several peaks, none wins):", rejected)
            if rejected < cases.STRUCTURED_MIN_REJECTIONS.get(case, 0):
                continue
            off = opeaks.upsampling_offset(*net.trained_net_output_resolution())
            prs = np.random.RandomState(3)
            ok = True
            for trial in range(4):                                          # decidability at the tolerance the test demands
                pert = maps + prs.uniform(-1e-4, 1e-4, maps.shape).astype(np.float32)
                pk = opeaks.keypoints_from_belief_maps(pert, off)
                ok = ok and np.array_equal(pk[..., 0] > -999, det) and (not det.any() or np.abs(pk - kps)[det].max() < 1e-3)
            if ok:
                break
        else:
            raise AssertionError("no channel mix gave a balanced fixture whose decisions are stable under +-1e-4")
        np.savez_compressed(os.path.join(HERE, "structured_%s.npz" % case), maps=maps, keypoints=kps, centres=centres,
                            final_weight=w[wk].numpy(), final_bias=w[bk].numpy())


def grad_sample(t, n=64):
    """Strided sample of a tensor (the same rule on both sides of the comparison)."""
    f = t.detach().flatten()
    return f[:: max(1, f.numel() // n)][:n].double().numpy().copy()


def resnet_training(dream):
    """G12: one DreamNetwork.train() step (dream/network.py:328-364) of the ResNet networks -- train-mode BatchNorm, the
    transposed-conv decoder, SGD -- run by the reference: loss, every parameter's gradient norm and a strided gradient sample,
    the updated parameters' samples and the BatchNorm running statistics after the step."""
    for case, (arch, manip, n_kp, (b, h, wd), final_keys) in cases.RESNET_TRAIN_CASES.items():
        cfg = ref_import.network_config(arch, manip, lr=cases.RESNET_TRAIN_LR, optimizer="sgd")
        cfg["training"]["config"]["net_input_resolution"] = [wd, h]
        net = dream.create_network_from_config_data(cfg)
        sd = net.model.state_dict()
        w = omodels.recipe_weights({key[len("module."):]: v for key, v in sd.items()}, final_keys=final_keys,
                                   final_scale=cases.TRAIN_FINAL_SCALE)
        net.model.load_state_dict({"module." + key: v for key, v in w.items()})
        net.enable_training()
        ow, oh = net.trained_net_output_resolution()
        x = torch.from_numpy(cases.image_batch(b, h, wd, seed=17))
        tgt = torch.from_numpy(cases.target_batch(b, n_kp, (ow, oh), in_wh=(wd, h), seed=17))
        loss = net.train([x], tgt)
        out = {"loss": np.array(loss.item(), np.float64)}
        for key, p in net.model.named_parameters():
            out["gradnorm/" + key] = np.array(float(p.grad.double().norm()))
            out["gradsample/" + key] = grad_sample(p.grad)
            out["param_sample/" + key] = grad_sample(p)
        for key, buf in net.model.named_buffers():
            if key.endswith("running_mean") or key.endswith("running_var"):
                out["buffer_sample/" + key] = grad_sample(buf)
        np.savez_compressed(os.path.join(HERE, "train_%s.npz" % case), **out)
        print("train", case, (b, h, wd), "->", (ow, oh), "loss", loss.item(), "keys", len(out))


def main():
    dream = ref_import.import_reference()
    torch.manual_seed(0)
    if "--only-resnet-training" in sys.argv:
        return resnet_training(dream)
    if "--only-peak-rules" in sys.argv:
        return peak_rules(dream)
    if "--only-structured" in sys.argv:
        return structured(dream)
    if "--only-belief-maps" in sys.argv:
        return belief_maps(dream)
    if "--only-conversions" in sys.argv:
        return keypoint_conversions(dream)
    if "--only-variants" in sys.argv:
        return variants(dream)
    if "--only-api" in sys.argv:
        return api_surface(dream)

    # ---- G1 + G2 -------------------------------------------------------------------------
    out = {}
    net_q = dream.create_network_from_config_data(ref_import.network_config("vgg_q"))
    for name, (maps, off) in cases.peak_cases().items():
        pk = dream.image_proc.peaks_from_belief_maps(torch.from_numpy(maps), off)
        c, xy, sc, ids = flatten_peaks(pk)
        out[name + "/counts"], out[name + "/xy"], out[name + "/score"], out[name + "/id"] = c, xy, sc, ids
        # selection rule through the reference's own DreamNetwork.inference: the model is replaced
        # by a stand-in returning these maps; the trained output resolution decides the offset.
        net_q.model = lambda x, _m=maps: [torch.from_numpy(_m)[None]]
        net_q.network_config["training"]["config"]["net_output_resolution"] = (
            [400, 400] if off == 0.0 else [100, 100])
        _, kps = net_q.inference(torch.zeros(1, 3, 8, 8))
        assert kps.dtype == torch.float32 and kps.device.type == "cpu"
        out[name + "/keypoints"] = kps.numpy()
    np.savez_compressed(os.path.join(HERE, "peaks_golden.npz"), **out)
    print("peaks_golden:", {k: v.shape for k, v in out.items() if k.endswith("counts")})

    # ---- G3 --------------------------------------------------------------------------------
    out = {}
    for name, (maps, beta) in cases.softargmax_cases().items():
        sm = dream.spatial_softmax.SoftArgmaxPavlo(n_keypoints=maps.shape[1], learned_beta=False,
                                                   initial_beta=beta)
        out[name] = sm(torch.from_numpy(maps)).numpy()
    np.savez_compressed(os.path.join(HERE, "softargmax_golden.npz"), **out)
    print("softargmax kat:", out["kat_b25"])

    # ---- G4 + G6 ---------------------------------------------------------------------------
    manifest = {}
    for arch, (k, manip, shapes) in cases.CNN_CASES.items():
        net = dream.create_network_from_config_data(ref_import.network_config(arch, manip))
        assert net.n_keypoints == k
        sd = net.model.state_dict()
        manifest[arch] = {key: list(v.shape) for key, v in sd.items()}
        w = omodels.recipe_weights({key[len("module."):]: v for key, v in sd.items()})
        net.model.load_state_dict({"module." + key: v for key, v in w.items()})
        net.enable_evaluation()
        out = {}
        for (b, h, wd) in shapes:
            x = torch.from_numpy(cases.image_batch(b, h, wd, seed=b * 1000 + h))
            with torch.no_grad():
                maps, kps = net.inference(x)
            tag = "%dx%dx%d" % (b, h, wd)
            y = maps.numpy()
            if y.size <= 200000:
                out[tag + "/maps"] = y
            else:   # keep the file small: strided sample + checksums
                out[tag + "/maps_sample"] = y[:, :, ::7, ::7].copy()
                out[tag + "/maps_sum"] = np.array([y.astype(np.float64).sum(), np.abs(y).astype(np.float64).sum()])
            out[tag + "/keypoints"] = kps.numpy()
            print(arch, tag, "->", y.shape, "absmax %.3f" % np.abs(y).max(),
                  "detections", int((kps.numpy()[..., 0] > -999).sum()), "/", kps.numpy().shape[0] * k)
        np.savez_compressed(os.path.join(HERE, "cnn_%s.npz" % arch), **out)
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=False)

    # ---- G5: one training step, vgg_q (config 3 at small resolution) ------------------------
    for opt in ("adam", "sgd"):
        cfg = ref_import.network_config("vgg_q", lr=cases.TRAIN_LR[opt], optimizer=opt)
        cfg["training"]["config"]["net_input_resolution"] = [96, 64]
        net = dream.create_network_from_config_data(cfg)
        sd = net.model.state_dict()
        w = omodels.recipe_weights({key[len("module."):]: v for key, v in sd.items()},
                                   final_keys=cases.TRAIN_FINAL_KEYS, final_scale=cases.TRAIN_FINAL_SCALE)
        net.model.load_state_dict({"module." + key: v for key, v in w.items()})
        net.enable_training()
        x = torch.from_numpy(cases.image_batch(2, 64, 96, seed=5))
        tgt = torch.from_numpy(cases.target_batch(2, 7, (24, 16), in_wh=(96, 64), seed=5))
        losses = []
        for step in range(3):
            loss = net.train([x], tgt)
            losses.append(loss.item())
            if step == 0:
                gn = {key: float(p.grad.double().norm()) for key, p in net.model.named_parameters()}
        out = {"losses": np.array(losses, np.float64)}
        for key, v in gn.items():
            out["gradnorm/" + key] = np.array(v)
        for key, p in net.model.named_parameters():
            out["param_sample/" + key] = p.detach().flatten()[:: max(1, p.numel() // 64)][:64].numpy().copy()
        np.savez_compressed(os.path.join(HERE, "train_vgg_q_%s.npz" % opt), **out)
        print("train", opt, "losses", losses)
    variants(dream)
    api_surface(dream)
    keypoint_conversions(dream)
    belief_maps(dream)
    structured(dream)
    peak_rules(dream)
    resnet_training(dream)


if __name__ == "__main__":
    main()
