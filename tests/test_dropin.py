"""DEV-CONTAINER ONLY (skipped where /root/reference is absent, e.g. on the GPU box): dream_amd.dropin makes the UNMODIFIED
reference package import with its hot path rebound to dream_amd -- no edit to /root/reference/dream/__init__.py:1-9 -- and the
calls scripts/train_network.py makes on it (:403-408 create / enable_training, :505 train, :507 .item(), :570 loss under
no_grad, :612-659 save_network, dream/analysis.py:147-149,210 load_state_dict / enable_evaluation / inference) run on the
dream_amd kernels (here: under the SIMT emulator, in a fresh interpreter so that no earlier import of the reference interferes)."""
import os
import subprocess
import sys

import pytest

import ref_import

HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPT = r'''
import os, sys, tempfile
sys.path[:0] = [%(root)r, %(tests)r, os.path.join(%(tests)r, "golden")]
import ref_import
ref_import.register_stubs()                       # third-party stand-ins only; the reference itself is NOT imported yet
import dream_amd.dropin                           # the one line a launcher / sitecustomize adds
import dream                                      # /root/reference/dream, its own __init__.py
import dream_amd, torch, numpy as np, cases
from emu_util import emulated_hip

assert dream.__file__.startswith("/root/reference/") and dream.__version__ == "1.3.0"
assert dream.network is dream_amd.network and dream.models is dream_amd.models
assert dream.DreamNetwork is dream_amd.network.DreamNetwork and dream.ResnetSimple is dream_amd.models.ResnetSimple
assert dream.create_network_from_config_data is dream_amd.network.create_network_from_config_data
assert dream.image_proc.__file__.startswith("/root/reference/")          # the reference's own module ...
assert dream.image_proc.peaks_from_belief_maps is dream_amd.image_proc.peaks_from_belief_maps   # ... with the HIP peak stage
assert dream.peaks_from_belief_maps is dream_amd.image_proc.peaks_from_belief_maps
assert dream.analysis.__file__.startswith("/root/reference/") and dream.KNOWN_OPTIMIZERS == ["adam", "sgd"]
assert "/root/reference/dream/network.py" not in [getattr(m, "__file__", None) for m in list(sys.modules.values())]

cfg = ref_import.network_config("vgg_q", lr=1e-5)                 # the dict scripts/train_network.py:259-323 assembles
cfg["training"]["config"]["net_input_resolution"] = [32, 32]
with emulated_hip():
    net = dream.create_network_from_config_data(cfg)              # train_network.py:403
    assert type(net).__module__ == "dream_amd.network"
    net.enable_training()                                         # :408
    x = torch.from_numpy(cases.image_batch(2, 32, 32, seed=1))
    t = torch.from_numpy(cases.target_batch(2, 7, (8, 8), in_wh=(32, 32), seed=1))
    losses = []
    for epoch in range(1):                                        # :464-514 epoch / batch loop
        net.enable_training()
        loss = net.train([x], t)                                  # :505
        losses.append(loss.item())                                # :507
        net.enable_evaluation()
        with torch.no_grad():
            vloss = net.loss([x], t).item()                       # :570
        out = tempfile.mkdtemp()
        net.save_network(out, "epoch_%%d" %% epoch, overwrite=True)    # :612-659
        assert os.path.exists(os.path.join(out, "epoch_%%d.pth" %% epoch))
    assert all(np.isfinite(losses)) and np.isfinite(vloss)
    net2 = dream.create_network_from_config_file(os.path.join(out, "epoch_0.yaml"))      # analysis.py:136-149
    net2.model.load_state_dict(torch.load(os.path.join(out, "epoch_0.pth")))
    net2.enable_evaluation()
    with torch.no_grad():
        maps, kps = net2.inference(x)                             # analysis.py:210
    assert tuple(kps.shape) == (2, 7, 2) and not kps.is_cuda and tuple(maps.shape) == (2, 7, 8, 8)
print("DROPIN-OK")
'''


@pytest.mark.skipif(not ref_import.have_reference(), reason="needs the reference checkout (dev container only)")
def test_unmodified_reference_package_runs_on_dream_amd():
    code = SCRIPT % {"root": os.path.dirname(HERE), "tests": HERE}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "DROPIN-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
