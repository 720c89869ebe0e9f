"""CPU: the single-process data-parallel executor behind ``training.platform.gpu_ids`` (dream_amd/data_parallel.py; the
reference's torch.nn.DataParallel at dream/network.py:244-256) with two EMULATED devices -- replicas on CPU tensors, kernels
under the SIMT emulator.  Checked: chunks are processed by different persistent replicas, inference results equal the
single-device ones bit for bit and in order, a training step equals the single-device step on the whole batch, the replicas are
refreshed after the optimizer step (one flat copy), the optimizer runs as ONE launch on the flat buffers."""
import os

import numpy as np
import pytest
import torch

import cases
import parity_checks as pc
from dream_amd import data_parallel, models, ops
from emu_util import emulated_hip
from oracle import models as om


@pytest.fixture
def emu():
    with emulated_hip() as lib:
        yield lib


@pytest.fixture
def two_devices(monkeypatch):
    monkeypatch.setenv("DREAM_DP_EMULATED_DEVICES", "2")


def _net(arch, lr=1e-6, opt="sgd", in_res=(32, 32), weights=None):
    if weights is None:
        weights = om.recipe_weights(om.build_model(arch, 7).state_dict(), cases.TRAIN_FINAL_KEYS, cases.TRAIN_FINAL_SCALE)
    return pc.build_network(arch, "cpu", weights=weights, optimizer=opt, lr=lr, in_res=in_res)


def test_parameters_live_in_one_flat_buffer_and_survive_load_state_dict():
    net = _net("vgg_q")
    rec = net.model.module._dream_flat
    assert data_parallel.flat_is_intact(net.model.module)
    assert rec["params"].numel() >= sum(p.numel() for p in net.model.parameters())
    sd = {k: v.clone() + 1 for k, v in net.model.state_dict().items()}
    net.model.load_state_dict(sd)                                    # in-place copies: the views stay where they are
    assert data_parallel.flat_is_intact(net.model.module)
    assert all(torch.equal(net.model.state_dict()[k], v) for k, v in sd.items())
    assert list(net.model.state_dict()) == ["module." + k for k in net.model.module.state_dict()]      # no replica keys


def test_inference_is_split_and_identical(emu, two_devices):
    net = _net("vgg_q")
    net.enable_evaluation()
    x = torch.from_numpy(cases.image_batch(3, 32, 32, seed=4))      # 3 frames over 2 devices: chunks of 2 and 1
    with torch.no_grad():
        maps, kps = net.inference(x)
    dp = net.model
    assert len(dp.devices()) == 2 and len(dp._replicas) == 1 and dp._replicas[0] is not dp.module
    assert data_parallel.flat_is_intact(dp._replicas[0])
    os.environ["DREAM_DP_EMULATED_DEVICES"] = "0"
    single = _net("vgg_q")
    single.enable_evaluation()
    with torch.no_grad():
        maps1, kps1 = single.inference(x)
    assert len(single.model.devices()) == 1
    assert torch.equal(maps, maps1) and torch.equal(kps, kps1) and kps.shape == (3, 7, 2)
    # a parameter change on the master reaches the replica before the next call (version-stamped flat copy)
    with torch.no_grad():
        for p in net.model.parameters():
            p.mul_(1.5)
        for p in single.model.parameters():
            p.mul_(1.5)
        m2, _ = net.inference(x)
        m2s, _ = single.inference(x)
    assert torch.equal(m2, m2s) and not torch.equal(m2, maps)


def test_training_step_equals_the_single_device_step(emu, two_devices, monkeypatch):
    x = torch.from_numpy(cases.image_batch(2, 32, 32, seed=9))
    t = torch.from_numpy(cases.target_batch(2, 7, (8, 8), in_wh=(32, 32), seed=9))
    launches = []
    real = ops.adam_step_
    monkeypatch.setattr(ops, "adam_step_", lambda *a, **k: (launches.append(a[0].numel()), real(*a, **k))[1])
    net = _net("vgg_q", lr=1e-5, opt="adam")
    net.enable_training()
    losses = [net.train([x], t).item() for _ in range(1)]
    assert len(net.model._replicas) == 1
    # the gradients handed to autograd are views of one flat buffer laid out like the parameters: one Adam launch per step
    grads = [p.grad for p in net.model.parameters()]
    base = grads[0].untyped_storage().data_ptr()
    assert all(g.untyped_storage().data_ptr() == base for g in grads)
    n_param = sum(p.numel() for p in net.model.parameters())
    assert len(launches) == 1 and n_param <= launches[0] <= net.model.module._dream_flat["params"].numel()
    # the replica was refreshed after each step
    rep = net.model._replicas[0]
    for (k, a), (_, b) in zip(net.model.module.named_parameters(), rep.named_parameters()):
        assert a.data_ptr() != b.data_ptr()
    net.model._sync_replicas(2)
    for (k, a), (_, b) in zip(net.model.module.named_parameters(), rep.named_parameters()):
        assert torch.equal(a, b), k
    os.environ["DREAM_DP_EMULATED_DEVICES"] = "0"
    del launches[:]
    single = _net("vgg_q", lr=1e-5, opt="adam")
    single.enable_training()
    losses1 = [single.train([x], t).item() for _ in range(1)]
    assert len(launches) == 1                                        # single device: one launch per step as well
    assert np.allclose(losses, losses1, rtol=2e-6), (losses, losses1)
    for (k, a), (_, b) in zip(net.model.named_parameters(), single.model.named_parameters()):
        assert float((a - b).abs().max()) <= 1e-7 + 1e-5 * float(b.abs().max()), k


def test_resnet_eval_replicas_follow_the_master_buffers(emu, two_devices):
    wts = om.recipe_weights(om.build_model("resnet_h", 7).state_dict())
    net = pc.build_network("resnet_h", "cpu", weights=wts, in_res=(32, 32))
    net.enable_evaluation()
    x = torch.from_numpy(cases.image_batch(2, 32, 32, seed=2))
    with torch.no_grad():
        maps, kps = net.inference(x)
    os.environ["DREAM_DP_EMULATED_DEVICES"] = "0"
    single = pc.build_network("resnet_h", "cpu", weights=wts, in_res=(32, 32))
    single.enable_evaluation()
    with torch.no_grad():
        maps1, kps1 = single.inference(x)
    assert torch.equal(maps, maps1) and torch.equal(kps, kps1)
    rep = net.model._replicas[0]
    assert not rep.training and torch.equal(rep.bn1.running_var, net.model.module.bn1.running_var)


def test_device_resolution_rules(monkeypatch):
    net = _net("vgg_q")
    assert net.model.device_ids is None and len(net.model.devices()) == 1          # CPU construction: pass-through
    dp = models.DreamDataParallel(torch.nn.Linear(2, 2), device_ids=[3, 1])
    assert dp.device_ids == [3, 1]                                                  # kept as given (reference attribute)
    assert dp.n_devices(1) == 1
