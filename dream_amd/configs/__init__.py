"""Architecture / manipulator presets (the reference's arch_configs/*.yaml and manip_configs/*.yaml
schema, SURVEY.md section 5 "Config / flags") and the merge that scripts/train_network.py:259-323
performs to build the ``network_config`` dict DreamNetwork consumes."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ARCHS = ("vgg_q", "vgg_f", "resnet_h", "resnet_f")
MANIPULATORS = ("panda", "kuka", "baxter")


def arch_config_path(arch):
    return os.path.join(_HERE, "arch", "dream_%s.yaml" % arch)


def manip_config_path(manip):
    return os.path.join(_HERE, "manip", "%s.yaml" % manip)


def default_network_config(arch="vgg_q", manip="panda", optimizer="adam", learning_rate=1e-4, batch_size=128,
                           gpu_ids=None):
    from ..network import _load_yaml
    if gpu_ids is None:
        # This helper builds a ONE-GPU configuration unless told otherwise: an empty ``gpu_ids`` means "every visible GPU" to
        # DreamNetwork (train_network.py:755-762, "Nothing specified means all GPUs"), which on an 8-GPU node would silently turn
        # every test / smoke / tool built on this helper into an 8-replica data-parallel run.  Pass ``gpu_ids=[]`` for that.
        import torch
        gpu_ids_cfg = [torch.cuda.current_device()] if torch.cuda.is_available() else []
    else:
        gpu_ids_cfg = [int(i) for i in gpu_ids]
    a = _load_yaml(arch_config_path(arch))
    m = _load_yaml(manip_config_path(manip))
    architecture = dict(a["architecture"])
    # train_network.py:245-251 copies the preprocessing choice into the architecture block
    architecture["image_preprocessing"] = a["training"]["config"]["image_preprocessing"]
    return {
        "data_path": "synthetic",
        "manipulator": m["manipulator"],
        "architecture": architecture,
        "training": {
            "config": {
                "epochs": 1,
                "batch_size": batch_size,
                "optimizer": {"type": optimizer, "learning_rate": learning_rate},
                "image_preprocessing": a["training"]["config"]["image_preprocessing"],
                "net_input_resolution": list(a["training"]["config"]["net_input_resolution"]),
            },
            "platform": {"gpu_ids": gpu_ids_cfg},
        },
    }
