"""Import the real reference (``/root/reference/dream``) in the GPU-less dev container.

DEV-CONTAINER ONLY: /root/reference does not exist on the GPU box, so nothing in the test suite
imports this module at run time except behind ``have_reference()``.  It exists so that
``make_golden.py`` can run the reference's own Python and commit its *outputs* as fixtures.

The reference needs third-party modules that are absent here (torchvision, cv2, ruamel.yaml,
albumentations, pyrr, webcolors) and calls ``.cuda()`` unconditionally.  We register stand-in
modules in ``sys.modules`` for the imports (only ``torchvision.models.vgg19/resnet101`` are ever
*called* on the hot path; they return the topology restated in oracle/topology.py -- that part is
"parity unpinned") and make ``.cuda()`` the identity.  The arithmetic that runs
is the reference's own code on torch's CPU kernels.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "dream"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns the imported reference package ``dream`` (version 1.3.0)."""
    if "dream" in sys.modules and getattr(sys.modules["dream"], "__version__", None) == "1.3.0":
        return sys.modules["dream"]
    register_stubs()
    import dream
    assert dream.__version__ == "1.3.0"
    return dream


def register_stubs():
    """Stand-ins for the reference's absent third-party imports, ``.cuda()`` as the identity, /root/reference on sys.path --
    everything import_reference() needs short of importing ``dream`` itself (tests/test_dropin.py imports it through
    dream_amd.dropin instead)."""
    assert have_reference(), "reference checkout not present (expected on the GPU box)"
    import torch
    import yaml
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import topology

    class _Vgg:
        def __init__(self):
            self.features = topology.vgg19_features()

    tv = _module("torchvision")
    tv.models = _module("torchvision.models",
                        vgg19=lambda pretrained=False, **kw: _Vgg(),
                        resnet101=lambda pretrained=False, **kw: topology.ResNet101())
    tv.transforms = _module("torchvision.transforms")
    tv.transforms.functional = _module("torchvision.transforms.functional")
    _module("cv2", SOLVEPNP_EPNP=1, SOLVEPNP_ITERATIVE=0)
    _module("albumentations")
    _module("webcolors")
    _module("pyrr", Quaternion=object)

    class _YAML:                       # PyYAML-backed stand-in for ruamel.yaml.YAML(typ="safe")
        def __init__(self, typ=None):
            pass

        def load(self, f):
            class _L(yaml.SafeLoader):
                pass
            _L.add_constructor("tag:yaml.org,2002:omap",
                               lambda ld, node: dict(kv for d in ld.construct_sequence(node, deep=True)
                                                     for kv in d.items()))
            return yaml.load(f, Loader=_L)

        def dump(self, data, f):
            yaml.safe_dump(data, f)

    ruamel = _module("ruamel")
    ruamel.yaml = _module("ruamel.yaml", YAML=_YAML)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def network_config(arch, manip="panda", n_keypoints=None, lr=1e-4, optimizer="adam", overrides=None):
    """The dict scripts/train_network.py:259-323 assembles from manip + arch YAML."""
    import yaml
    dream = import_reference()
    data_parser = sys.modules["ruamel.yaml"].YAML(typ="safe")
    with open(os.path.join(REFERENCE_ROOT, "arch_configs", "dream_%s.yaml" % arch)) as f:
        arch_cfg = data_parser.load(f)
    with open(os.path.join(REFERENCE_ROOT, "manip_configs", "%s.yaml" % manip)) as f:
        manip_cfg = data_parser.load(f)
    architecture = dict(arch_cfg["architecture"])
    architecture.update(overrides or {})
    architecture["image_preprocessing"] = arch_cfg["training"]["config"]["image_preprocessing"]
    return {
        "data_path": "synthetic",
        "manipulator": manip_cfg["manipulator"],
        "architecture": architecture,
        "training": {
            "config": {
                "epochs": 1, "training_data_fraction": 0.8, "validation_data_fraction": 0.2,
                "batch_size": 2, "data_augmentation": {"image_rgb": False},
                "worker_size": 0,
                "optimizer": {"type": optimizer, "learning_rate": lr},
                "image_preprocessing": arch_cfg["training"]["config"]["image_preprocessing"],
                "net_input_resolution": list(arch_cfg["training"]["config"]["net_input_resolution"]),
            },
            "platform": {"user": "golden", "hostname": "devbox", "gpu_ids": []},
        },
    }
