"""CPU: pin oracle.models against outputs of the stub-imported reference (tests/golden/cnn_*.npz,
train_vgg_q_*.npz, state_dict_manifest.json) and the released-checkpoint sizes."""
import json
import os

import numpy as np
import pytest
import torch

import cases
from oracle import models as om
from oracle import peaks as op

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("arch", ["vgg_q", "vgg_f", "resnet_h", "resnet_f"])
def test_state_dict_manifest_and_checkpoint_size(arch):
    k = cases.CNN_CASES[arch][0]
    sd = om.build_model(arch, k).state_dict()
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))[arch]
    assert ["module." + key for key in sd.keys()] == list(man.keys())
    assert all(list(sd[key[len("module."):]].shape) == shp for key, shp in man.items())
    nparam = sum(v.numel() for key, v in sd.items() if v.dtype.is_floating_point and "running" not in key)
    # trained_models/DOWNLOAD.sh:12-38 : 85 / 86 / 207 / 211 MB checkpoints (K=7 for the shipped ones)
    expect = {"vgg_q": 22220615, "vgg_f": 22442055, "resnet_h": 54039367, "resnet_f": 55091281}[arch]
    assert nparam == expect


@pytest.mark.parametrize("arch", ["vgg_q", "vgg_f", "resnet_h", "resnet_f"])
def test_cnn_matches_reference_outputs(arch):
    k, _, shapes = cases.CNN_CASES[arch]
    g = np.load(os.path.join(GOLD, "cnn_%s.npz" % arch))
    m = om.build_model(arch, k)
    m.load_state_dict(om.recipe_weights(m.state_dict()))
    m.eval()
    for (b, h, w) in shapes:
        if h * w > 100000 and os.environ.get("DREAM_FULL_ORACLE", "0") != "1" and arch != "vgg_q":
            continue
        x = torch.from_numpy(cases.image_batch(b, h, w, seed=b * 1000 + h))
        with torch.no_grad():
            y = m(x)[0].numpy()
        tag = "%dx%dx%d" % (b, h, w)
        if tag + "/maps" in g:
            ref = g[tag + "/maps"]
            err = np.abs(y - ref).max()
        else:
            ref = g[tag + "/maps_sample"]
            err = np.abs(y[:, :, ::7, ::7] - ref).max()
        # same torch CPU kernels; thread-count dependent summation order only
        assert err <= 1e-5 * max(1.0, np.abs(ref).max()), (arch, tag, err)
        off = op.upsampling_offset(*{"vgg_q": (100, 100), "vgg_f": (400, 400), "resnet_h": (208, 208),
                                     "resnet_f": (416, 416)}[arch])
        kps = op.keypoints_from_belief_maps(y, off)
        ref_k = g[tag + "/keypoints"]
        assert np.array_equal(kps == np.float32(-999.999), ref_k == np.float32(-999.999))
        assert np.abs(kps - ref_k).max() < 1e-3


@pytest.mark.parametrize("arch", sorted(cases.STRUCTURED_CASES))
def test_structured_fixture_regenerates_from_the_oracle(arch):
    """The structured (blob-like, magnitude-1) fixture: weights and frames regenerate from the committed recipes and the
    oracle reproduces the reference's maps / keypoints -- on any box, without the reference."""
    case = arch
    arch, manip, k, last, (b, h, w), recipe, zero_bg = cases.STRUCTURED_CASES[case]
    g = np.load(os.path.join(GOLD, "structured_%s.npz" % case))
    m = om.build_model(arch, k)
    wts = {"structured": om.structured_weights, "smooth": om.smooth_weights, "recipe": om.recipe_weights}[recipe](m.state_dict())
    wts[last + ".weight"], wts[last + ".bias"] = torch.from_numpy(g["final_weight"]), torch.from_numpy(g["final_bias"])
    m.load_state_dict(wts)
    m.eval()
    x, centres = cases.structured_input(case)
    assert np.array_equal(centres, g["centres"])
    with torch.no_grad():
        y = m(torch.from_numpy(x))[0].numpy()
    # magnitude 1: the strongest response is scaled to 1 (smooth recipe: the strongest POSITIVE one; its maps dip to -6 where a blob's colour
    # clamps the first layer's ReLU -- the absolute 1e-4 bound of the GPU test is the harder for it)
    top = g["maps"].max() if recipe == "smooth" else np.abs(g["maps"]).max()
    assert np.abs(y - g["maps"]).max() <= 1e-5 and 0.99 <= top <= 1.0 + 1e-6 and np.abs(g["maps"]).max() <= 8.0
    kps = op.keypoints_from_belief_maps(y, op.upsampling_offset(y.shape[3], y.shape[2]))
    det = g["keypoints"][..., 0] > -999
    want_detections = cases.STRUCTURED_MIN_DETECTIONS.get(case, 0.25) > 0      # (vgg_f_recipe is kept for its map values: all rejected)
    assert np.array_equal(kps[..., 0] > -999, det) and (0 if want_detections else -1) < det.sum() < det.size
    assert not det.any() or np.abs(kps - g["keypoints"])[det].max() < 1e-3


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_train_step_matches_reference(opt):
    g = np.load(os.path.join(GOLD, "train_vgg_q_%s.npz" % opt))
    m = om.build_model("vgg_q", 7)
    m.load_state_dict(om.recipe_weights(m.state_dict(), cases.TRAIN_FINAL_KEYS, cases.TRAIN_FINAL_SCALE))
    m.train()
    lr = cases.TRAIN_LR[opt]
    o = torch.optim.Adam(m.parameters(), lr=lr) if opt == "adam" else torch.optim.SGD(m.parameters(), lr=lr)
    x = torch.from_numpy(cases.image_batch(2, 64, 96, seed=5))
    t = torch.from_numpy(cases.target_batch(2, 7, (24, 16), in_wh=(96, 64), seed=5))
    losses = []
    for step in range(3):
        o.zero_grad()
        loss = torch.nn.functional.mse_loss(m(x)[0], t)
        loss.backward()
        o.step()
        losses.append(loss.item())
        if step == 0:
            for key, p in m.named_parameters():
                ref = float(g["gradnorm/module." + key])
                assert abs(float(p.grad.double().norm()) - ref) <= 1e-4 * max(ref, 1e-6), key
    assert np.allclose(losses, g["losses"], rtol=1e-4)
    for key, p in m.named_parameters():
        s = p.detach().flatten()[:: max(1, p.numel() // 64)][:64].numpy()
        assert np.allclose(s, g["param_sample/module." + key], rtol=1e-4, atol=1e-6), key


def test_softargmax_matches_reference_outputs():
    g = np.load(os.path.join(GOLD, "softargmax_golden.npz"))
    for name, (maps, beta) in cases.softargmax_cases().items():
        sm = om.SoftArgmaxPavlo(maps.shape[1], learned_beta=False, initial_beta=beta)
        y = sm(torch.from_numpy(maps)).numpy()
        assert np.allclose(y, g[name], rtol=0, atol=1e-4), name
    # SURVEY.md 8a6 KAT (border bias from the zero-padded avg-pool is expected behaviour)
    assert np.allclose(g["kat_b25"][0], [[7.5230, 5.3138], [14.9651, 9.9651], [25.0729, 16.0064]], atol=1e-3)


@pytest.mark.parametrize("name", sorted(cases.VARIANT_CASES))
def test_variant_hourglasses_match_reference(name):
    """Skip connections, full-resolution upsample decoder, soft-argmax head and DreamHourglassMultiStage: state_dict
    layout, every head / stage output and two Adam steps against the stub-imported reference (variant_*.npz)."""
    shapes, train = cases.VARIANT_CASES[name]
    g = np.load(os.path.join(GOLD, "variant_%s.npz" % name))
    man = json.load(open(os.path.join(GOLD, "variant_state_dict_manifest.json")))[name]
    m = om.build_model(name, 7)
    sd = m.state_dict()
    assert ["module." + key for key in sd.keys()] == list(man.keys())
    assert all(list(sd[key[len("module."):]].shape) == shp for key, shp in man.items())
    m.load_state_dict(om.recipe_weights(sd))
    m.eval()
    for (b, h, w) in shapes:
        tag = "%dx%dx%d" % (b, h, w)
        x = torch.from_numpy(cases.image_batch(b, h, w, seed=b * 1000 + h))
        with torch.no_grad():
            heads = m(x)
        for i, t in enumerate(heads):
            ref = g[tag + "/head%d" % i]
            assert np.abs(t.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (name, tag, i)
    if train:
        b, h, w = cases.VARIANT_TRAIN_SHAPE
        m = om.build_model(name, 7)
        m.load_state_dict(om.recipe_weights(m.state_dict(), cases.TRAIN_FINAL_KEYS, cases.TRAIN_FINAL_SCALE))
        m.train()
        o = torch.optim.Adam(m.parameters(), lr=cases.TRAIN_LR["adam"])
        x = torch.from_numpy(cases.image_batch(b, h, w, seed=7))
        with torch.no_grad():
            oh, ow = m(x[:1])[0].shape[2:]
        t = torch.from_numpy(cases.target_batch(b, 7, (ow, oh), in_wh=(w, h), seed=7))
        losses = []
        for step in range(2):
            o.zero_grad()
            outs = m(x)
            if "n_stages" in om.VARIANTS[name][1]:           # network.py:345-352
                loss = torch.nn.functional.mse_loss(torch.stack(outs), t.unsqueeze(0).expand(len(outs), -1, -1, -1, -1))
            else:
                loss = torch.nn.functional.mse_loss(outs[0], t)
            loss.backward()
            o.step()
            losses.append(loss.item())
            if step == 0:
                for key, p in m.named_parameters():
                    ref = float(g["train/gradnorm/module." + key])
                    assert abs(float(p.grad.double().norm()) - ref) <= 1e-4 * max(ref, 1e-6), key
        assert np.allclose(losses, g["train/losses"], rtol=1e-5)


@pytest.mark.parametrize("case", sorted(cases.RESNET_TRAIN_CASES))
def test_resnet_train_step_matches_reference(case):
    """G12: the oracle's ResNet training step (train-mode BatchNorm, transposed-conv decoder, SGD) reproduces the reference's
    DreamNetwork.train() step: loss, every gradient norm / sample, updated parameters, running statistics."""
    arch, manip, k, (b, h, w), final_keys = cases.RESNET_TRAIN_CASES[case]
    g = np.load(os.path.join(GOLD, "train_%s.npz" % case))
    m = om.build_model(arch, k)
    m.load_state_dict(om.recipe_weights(m.state_dict(), final_keys, cases.TRAIN_FINAL_SCALE))
    m.train()
    o = torch.optim.SGD(m.parameters(), lr=cases.RESNET_TRAIN_LR)
    x = torch.from_numpy(cases.image_batch(b, h, w, seed=17))
    out = m(x)[0]
    t = torch.from_numpy(cases.target_batch(b, k, (out.shape[3], out.shape[2]), in_wh=(w, h), seed=17))
    o.zero_grad()
    loss = torch.nn.functional.mse_loss(out, t)
    loss.backward()
    o.step()
    assert abs(loss.item() - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))

    def sample(v):
        f = v.detach().flatten()
        return f[:: max(1, f.numel() // 64)][:64].double().numpy()
    for key, p in m.named_parameters():
        ref = float(g["gradnorm/module." + key])
        assert abs(float(p.grad.double().norm()) - ref) <= 1e-5 * max(ref, 1e-12), key
        assert np.abs(sample(p.grad) - g["gradsample/module." + key]).max() <= 1e-5 * max(np.abs(g["gradsample/module." + key]).max(), 1e-12), key
        assert np.abs(sample(p) - g["param_sample/module." + key]).max() <= 1e-6 * max(np.abs(g["param_sample/module." + key]).max(), 1e-12), key
    for key, buf in m.named_buffers():
        if key.endswith("running_mean") or key.endswith("running_var"):
            assert np.abs(sample(buf) - g["buffer_sample/module." + key]).max() <= 1e-6 * max(np.abs(g["buffer_sample/module." + key]).max(), 1e-12), key
