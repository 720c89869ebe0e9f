"""CPU: the ImageNet-initialisation hook (dream_amd/pretrained.py; reference dream/models.py:19-32,587-615) and the
checkpoint verification tool (tools/verify_checkpoint.py; reference dream/network.py:29-63,592-632)."""
import contextlib
import io
import os
import sys
import types
import warnings

import pytest
import torch

import dream_amd
from dream_amd import pretrained
from oracle import topology

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _fake_imagenet(arch):
    """A stand-in for torchvision's pretrained model: the restated topology with a recognisable value in every tensor."""
    torch.manual_seed(1234 if arch == "vgg19" else 4321)
    if arch == "vgg19":
        net = torch.nn.Module()
        net.features = topology.vgg19_features()
    else:
        net = topology.ResNet101()
    with torch.no_grad():
        for i, (k, v) in enumerate(net.state_dict().items()):
            if v.dtype.is_floating_point:
                v.copy_(torch.randn_like(v) * 0.01 + (i % 7))
    return net


@pytest.fixture
def fake_torchvision(monkeypatch):
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    made = {}

    def ctor(arch):
        def make(weights=None, pretrained=None, **kw):
            assert weights == "IMAGENET1K_V1" or pretrained is True
            made[arch] = _fake_imagenet(arch)
            return made[arch]
        return make
    tvm.vgg19, tvm.resnet101 = ctor("vgg19"), ctor("resnet101")
    tv.models = tvm
    monkeypatch.setitem(sys.modules, "torchvision", tv)
    monkeypatch.setitem(sys.modules, "torchvision.models", tvm)
    monkeypatch.setattr(pretrained, "_warned", set())
    return made


def test_hourglass_takes_the_vgg19_encoder_except_the_first_conv(fake_torchvision):
    with warnings.catch_warnings():
        warnings.simplefilter("error", pretrained.PretrainedUnavailable)
        m = dream_amd.models.DreamHourglass(7, internalize_spatial_softmax=False)
    assert m.imagenet_initialised
    tv_sd = fake_torchvision["vgg19"].state_dict()
    sd = m.state_dict()
    copied = 0
    for cname in ("layer_0_1_down", "layer_0_2_down", "layer_0_3_down", "layer_0_4_down", "layer_0_5_down"):
        for idx, _ in getattr(m, cname).named_children():
            for part in ("weight", "bias"):
                same = torch.equal(sd["%s.%s.%s" % (cname, idx, part)], tv_sd["features.%s.%s" % (idx, part)])
                assert same == (idx != "0"), (cname, idx)          # models.py:592-597: a fresh first conv
                copied += int(same)
    assert copied == 2 * len(pretrained.VGG19_REUSED)
    assert float(sd["heads_0.0.weight"].abs().max()) < 1.0          # decoder / head: default initialisation


def test_resnet_takes_the_resnet101_trunk_and_honours_pretrained(fake_torchvision):
    m = dream_amd.models.ResnetSimple(7)
    assert m.imagenet_initialised
    tv_sd, sd = fake_torchvision["resnet101"].state_dict(), m.state_dict()
    trunk = [k for k in sd if k.split(".")[0] in ("conv1", "bn1", "layer1", "layer2", "layer3", "layer4")]
    assert len(trunk) == 624 and all(torch.equal(sd[k], tv_sd[k]) for k in trunk)
    assert not any(k.startswith("upsample") and torch.equal(sd[k], tv_sd.get(k, torch.zeros(0))) for k in sd)
    fake_torchvision.clear()
    m2 = dream_amd.models.ResnetSimple(7, pretrained=False)
    assert not m2.imagenet_initialised and "resnet101" not in fake_torchvision       # torchvision not even asked


def test_weight_file_override_needs_no_torchvision(tmp_path, monkeypatch):
    monkeypatch.setattr(pretrained, "_warned", set())
    monkeypatch.setitem(sys.modules, "torchvision", None)             # import torchvision -> ImportError
    path = str(tmp_path / "vgg19.pth")
    ref = _fake_imagenet("vgg19")
    torch.save(ref.state_dict(), path)
    monkeypatch.setenv("DREAM_VGG19_WEIGHTS", path)
    m = dream_amd.models.DreamHourglass(7, internalize_spatial_softmax=False)
    assert m.imagenet_initialised
    assert torch.equal(getattr(m.layer_0_5_down, "34").weight, ref.state_dict()["features.34.weight"])


def test_missing_weights_warn_once_and_loudly(monkeypatch):
    monkeypatch.setattr(pretrained, "_warned", set())
    monkeypatch.setitem(sys.modules, "torchvision", None)
    monkeypatch.delenv("DREAM_VGG19_WEIGHTS", raising=False)
    monkeypatch.delenv("DREAM_RESNET101_WEIGHTS", raising=False)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        a = dream_amd.models.DreamHourglass(7, internalize_spatial_softmax=False)
        b = dream_amd.models.DreamHourglass(7, internalize_spatial_softmax=False)
        c = dream_amd.models.ResnetSimple(7)
    mine = [w for w in rec if issubclass(w.category, pretrained.PretrainedUnavailable)]
    assert not (a.imagenet_initialised or b.imagenet_initialised or c.imagenet_initialised)
    assert len(mine) == 2                                             # one per architecture, not one per construction
    assert "DIFFERENT POINT THAN THE REFERENCE" in str(mine[0].message) and "DREAM_VGG19_WEIGHTS" in str(mine[0].message)


# ---- tools/verify_checkpoint.py --------------------------------------------------------------------------------------
@pytest.mark.parametrize("arch", ["vgg_q", "resnet_h"])
def test_verify_checkpoint_accepts_a_saved_network_and_names_what_is_wrong(arch, tmp_path):
    import verify_checkpoint as vc
    net = _quiet(dream_amd.create_network_from_config_data, dream_amd.default_network_config(arch, "panda"))
    _quiet(net.save_network, str(tmp_path / "out"), "net")
    yaml_path, pth_path = str(tmp_path / "out" / "net.yaml"), str(tmp_path / "out" / "net.pth")
    lines = []
    assert _quiet(vc.verify, yaml_path, pth_path, out=lines.append) == 0
    assert any("manifest: keys, shapes and dtypes agree" in ln for ln in lines)
    assert any("bit-identical" in ln for ln in lines)
    # a checkpoint of another keypoint count / with a renamed key must be rejected with the offending keys listed
    sd = torch.load(pth_path)
    last = [k for k in sd if k.endswith("weight")][-1]
    broken = dict(sd)
    broken[last] = torch.zeros((9,) + tuple(sd[last].shape[1:]))
    broken["module.extra.weight"] = torch.zeros(1)
    del broken[[k for k in sd if k.endswith("bias")][0]]
    bad_path = str(tmp_path / "bad.pth")
    torch.save(broken, bad_path)
    lines = []
    assert _quiet(vc.verify, yaml_path, bad_path, out=lines.append) == 1
    text = "\n".join(lines)
    assert "shape mismatch: " + last in text and "unexpected in checkpoint: module.extra.weight" in text
    assert "missing in checkpoint" in text and "FAIL: 3 manifest problem(s)" in text
