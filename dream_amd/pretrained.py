"""ImageNet initialisation of the encoder / trunk, as the reference does it.

The reference never starts from random weights: ``DreamHourglass`` takes every encoder conv but the first from
``torchvision.models.vgg19(pretrained=True).features`` (/root/reference/dream/models.py:587-615) and ``ResnetSimple`` takes
``conv1, bn1, layer1..4`` from ``torchvision.models.resnet101(pretrained=pretrained)`` (:19-32).  torchvision is an
un-vendored dependency of the reference (requirements.txt:16) and its weight files are a download, so this module is a
guarded hook:

  1. ``DREAM_VGG19_WEIGHTS`` / ``DREAM_RESNET101_WEIGHTS`` = path of a torchvision ``state_dict`` file (``vgg19-dcbb9e9d.pth``,
     ``resnet101-*.pth``) -- for air-gapped machines, no torchvision needed;
  2. otherwise ``torchvision.models.vgg19 / resnet101`` with pretrained weights when torchvision is importable and the
     weights are in its cache (or downloadable);
  3. otherwise ONE loud warning per process and architecture: training from scratch then starts from PyTorch's default
     initialisation, which is NOT the reference's starting point.  Inference and ``load_state_dict`` of a trained
     checkpoint are unaffected (every tensor is overwritten).

Only tensors are copied (into the parameter containers of dream_amd.models, under the reference's state_dict keys); no
torchvision module ends up in the model.
"""
import os
import warnings

import torch

# torchvision vgg19.features indices of the convs the reference re-uses (models.py:598-615); index 0 is replaced by a fresh conv
VGG19_REUSED = (2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34)
_warned = set()


class PretrainedUnavailable(UserWarning):
    pass


def _warn_once(arch, why):
    if arch in _warned:
        return
    _warned.add(arch)
    warnings.warn(
        "dream_amd: ImageNet-pretrained %s weights are NOT available (%s). The reference always starts from them "
        "(dream/models.py:22,587); this model keeps PyTorch's default initialisation instead, so TRAINING FROM SCRATCH "
        "STARTS FROM A DIFFERENT POINT THAN THE REFERENCE. Loading a trained checkpoint is unaffected. To fix: install "
        "torchvision with its weight cache, or point DREAM_%s_WEIGHTS at a torchvision state_dict file."
        % (arch, why, arch.upper()), PretrainedUnavailable, stacklevel=3)


def _torchvision_state_dict(arch):
    """state_dict of torchvision's ImageNet model ``arch`` ("vgg19" | "resnet101") or (None, reason)."""
    path = os.environ.get("DREAM_%s_WEIGHTS" % arch.upper())
    if path:
        if not os.path.exists(path):
            return None, "DREAM_%s_WEIGHTS=%s does not exist" % (arch.upper(), path)
        sd = torch.load(path, map_location="cpu")
        return (sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()), None
    try:
        import torchvision.models as tvm
    except Exception as e:                                   # ImportError, or a broken install
        return None, "torchvision is not importable: %s" % (e,)
    ctor = getattr(tvm, arch, None)
    if ctor is None:
        return None, "torchvision.models has no %s" % arch
    try:
        try:
            net = ctor(weights="IMAGENET1K_V1")            # torchvision >= 0.13
        except TypeError:
            net = ctor(pretrained=True)                     # the call the reference makes
    except Exception as e:                                   # no cache and no network
        return None, "torchvision could not provide the weights: %s" % (e,)
    try:
        return net.state_dict(), None
    except Exception as e:                                   # a stand-in torchvision (test harnesses) without real modules
        return None, "torchvision.models.%s() did not return a module with a state_dict: %s" % (arch, e)


def init_vgg19_encoder(hourglass):
    """Copy vgg19.features[i].{weight,bias} for the re-used indices into ``hourglass.layer_0_k_down`` (models.py:598-615).
    Returns True when the weights were applied."""
    sd, why = _torchvision_state_dict("vgg19")
    if sd is None:
        _warn_once("vgg19", why)
        return False
    targets = {}
    for cname in ("layer_0_1_down", "layer_0_2_down", "layer_0_3_down", "layer_0_4_down", "layer_0_5_down"):
        for idx, mod in getattr(hourglass, cname).named_children():
            if int(idx) in VGG19_REUSED:
                targets[int(idx)] = mod
    assert sorted(targets) == sorted(VGG19_REUSED), "encoder containers do not match vgg19.features"
    with torch.no_grad():
        for idx, mod in targets.items():
            w, b = sd["features.%d.weight" % idx], sd["features.%d.bias" % idx]
            assert tuple(w.shape) == tuple(mod.weight.shape), (idx, tuple(w.shape), tuple(mod.weight.shape))
            mod.weight.copy_(w)
            mod.bias.copy_(b)
    return True


def init_resnet101_trunk(resnet):
    """Copy conv1 / bn1 / layer1..4 (parameters and BatchNorm buffers) of torchvision's resnet101 (models.py:22-32)."""
    sd, why = _torchvision_state_dict("resnet101")
    if sd is None:
        _warn_once("resnet101", why)
        return False
    own = resnet.state_dict()
    picked = {k: v for k, v in sd.items() if k.split(".")[0] in ("conv1", "bn1", "layer1", "layer2", "layer3", "layer4")}
    missing = [k for k in own if k.split(".")[0] in ("conv1", "bn1", "layer1", "layer2", "layer3", "layer4") and k not in picked]
    assert not missing, "torchvision resnet101 state_dict lacks %s" % missing[:3]
    with torch.no_grad():
        for k, v in picked.items():
            assert tuple(own[k].shape) == tuple(v.shape), (k, tuple(own[k].shape), tuple(v.shape))
            own[k].copy_(v)
    return True
