"""CPU: the single-process data-parallel executor behind ``training.platform.gpu_ids`` (dream_amd/data_parallel.py; the
reference's torch.nn.DataParallel at dream/network.py:244-256) with two EMULATED devices -- replicas on CPU tensors, kernels
under the SIMT emulator.  Checked: chunks are processed by different persistent replicas, inference results equal the
single-device ones bit for bit and in order, a training step equals the single-device step on the whole batch, the replicas are
kept identical by applying the optimizer's ONE flat launch on every replica to the all-reduced gradients (no parameter copy
after a step), a foreign optimizer is noticed and repaired by one flat copy, HipAdam resumes from a saved state."""
import copy
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
    # one launch on the master and the identical launch on the replica
    assert len(launches) == 2 and launches[0] == launches[1] and n_param <= launches[0] <= net.model.module._dream_flat["params"].numel()
    # the replica applied the same update to the same (all-reduced) gradients: identical without any copy from the master
    rep = net.model._replicas[0]
    copies = net.model.stats["param_copies"]
    assert copies == 1 and net.model.stats["replica_steps"] == 1            # the one copy: creation of the replica
    for (k, a), (_, b) in zip(net.model.module.named_parameters(), rep.named_parameters()):
        assert a.data_ptr() != b.data_ptr()
        assert torch.equal(a, b), k
    assert torch.equal(net.model.module._dream_flat["grads"], rep._dream_flat["grads"])
    net.model._sync_replicas(2)
    assert net.model.stats["param_copies"] == copies
    os.environ["DREAM_DP_EMULATED_DEVICES"] = "0"
    del launches[:]
    single = _net("vgg_q", lr=1e-5, opt="adam")
    single.enable_training()
    losses1 = [single.train([x], t).item() for _ in range(1)]
    assert len(launches) == 1                                        # single device: one launch per step as well
    assert np.allclose(losses, losses1, rtol=2e-6), (losses, losses1)
    for (k, a), (_, b) in zip(net.model.named_parameters(), single.model.named_parameters()):
        assert float((a - b).abs().max()) <= 1e-7 + 1e-5 * float(b.abs().max()), k
    # anything else that changes the master (an optimizer that is not dream_amd's, load_state_dict, a manual edit) moves the
    # version stamp: the replicas are then repaired by one flat copy, and their optimizer state is dropped with them
    with torch.no_grad():
        next(net.model.module.parameters()).add_(0.5)
    assert not torch.equal(rep._dream_flat["params"], net.model.module._dream_flat["params"])
    net.model._sync_replicas(2)
    assert net.model.stats["param_copies"] == copies + 1 and not net.model._opt_state
    assert torch.equal(rep._dream_flat["params"], net.model.module._dream_flat["params"])


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


def test_four_replicas_stay_identical_over_steps_and_match_one_device(emu, monkeypatch):
    monkeypatch.setenv("DREAM_DP_EMULATED_DEVICES", "4")
    x = torch.from_numpy(cases.image_batch(4, 32, 32, seed=5))
    t = torch.from_numpy(cases.target_batch(4, 7, (8, 8), in_wh=(32, 32), seed=5))
    net = _net("vgg_q", lr=1e-5, opt="adam")
    net.enable_training()
    reduced = []
    real = ops.allreduce_sum_
    monkeypatch.setattr(ops, "allreduce_sum_", lambda flats: (reduced.append(len(flats)), real(flats))[1])
    losses = [net.train([x], t).item() for _ in range(1)]
    dp = net.model
    assert len(dp._replicas) == 3 and reduced == [4]                    # ONE all-reduce over the four flat buffers per step
    assert dp.stats["param_copies"] == 3 and dp.stats["replica_steps"] == 3
    for rep in dp._replicas:
        assert torch.equal(rep._dream_flat["params"], dp.module._dream_flat["params"])
    os.environ["DREAM_DP_EMULATED_DEVICES"] = "0"
    single = _net("vgg_q", lr=1e-5, opt="adam")
    single.enable_training()
    losses1 = [single.train([x], t).item() for _ in range(1)]
    assert np.allclose(losses, losses1, rtol=2e-6), (losses, losses1)
    for (k, a), (_, b) in zip(net.model.named_parameters(), single.model.named_parameters()):
        assert float((a - b).abs().max()) <= 1e-7 + 2e-5 * float(b.abs().max()), k


def test_adam_resumes_from_a_saved_state_bit_for_bit(emu):
    """SURVEY.md 8f rank 4 (resume incl. optimizer state): HipAdam on parameters that are views of one flat buffer (the layout
    of every DreamNetwork model) -- two continuous steps == one step, optimizer.state_dict() -> a fresh model + optimizer ->
    load_state_dict, one more step, bit for bit; the loaded moments are adopted into the flat moment buffers (round 2 dropped them)."""
    from dream_amd.optim import HipAdam

    def model():
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5))
        data_parallel.flatten_module_(m)
        return m

    def step(m, opt, seed):
        g = torch.Generator().manual_seed(seed)
        for p in m.parameters():
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()

    a = model()
    oa = HipAdam(list(a.parameters()), lr=1e-2)
    step(a, oa, 1)
    model_sd = {k: v.clone() for k, v in a.state_dict().items()}
    opt_sd = copy.deepcopy(oa.state_dict())
    step(a, oa, 2)
    b = model()
    b.load_state_dict(model_sd)
    ob = HipAdam(list(b.parameters()), lr=1e-2)
    ob.load_state_dict(opt_sd)
    step(b, ob, 2)
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa, pb), k
    st = ob.state[next(iter(b.parameters()))]
    assert int(st["step"]) == 2 and st["exp_avg"].abs().sum() > 0 and "moments" in ob._plan
    fresh = model()
    fresh.load_state_dict(model_sd)
    of = HipAdam(list(fresh.parameters()), lr=1e-2)                   # the bug of round 2: moments and step restart from zero
    step(fresh, of, 2)
    assert not torch.equal(next(iter(fresh.parameters())), next(iter(a.parameters())))


# ---- gradients that outlive a step (round-3 advisor findings) -----------------------------------------------------------------
# The executor's bookkeeping does not depend on the network: these tests drive DreamDataParallel with a two-tensor module that
# implements the replica protocol (dp_forward / dp_backward / ...) in plain torch, so that they take seconds under the emulator
# (a vgg_q replica step is ~10 s of emulated MFMAs); the optimizers and the all-reduce are the product's (emulated kernels).
class _TinyReplica(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(7)
        self.w = torch.nn.Parameter(torch.randn(5, 12, generator=g))
        self.b = torch.nn.Parameter(torch.randn(5, generator=g))

    def dp_parameters(self):
        return [self.w, self.b]

    def dp_trainable(self):
        return True

    def dp_forward(self, x, save):
        f = x.flatten(1)[:, :12]
        return [f @ self.w.t() + self.b], (f if save else None)

    def dp_backward(self, f, grad_outs, reducer=None):
        g = grad_outs[0]
        return [g.t() @ f, g.sum(0)]

    def dp_finish(self, outs):
        return outs

    def forward(self, x):
        return [x.flatten(1)[:, :12] @ self.w.t() + self.b]


def _tiny_dp(opt="sgd", lr=1e-2):
    from dream_amd.optim import HipAdam, HipSGD, attach_data_parallel
    mod = _TinyReplica()
    dp = models.DreamDataParallel(mod)
    dp.flatten_parameters()
    params = list(dp.parameters())
    optim = HipAdam(params, lr=lr) if opt == "adam" else HipSGD(params, lr=lr)
    attach_data_parallel(optim, dp)
    dp.train()
    return dp, optim


def _tiny_batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, 2, 2, generator=g), torch.randn(n, 5, generator=g)


def _tiny_loss(dp, x, t):
    return ((dp(x)[0] - t) ** 2).mean()


def _grads(dp):
    return [p.grad.clone() for p in dp.parameters()]


def _tiny_reference(x, t):
    ref = _TinyReplica()
    ((ref(x)[0] - t) ** 2).mean().backward()
    return [ref.w.grad, ref.b.grad]


def test_gradient_accumulation_and_zero_grad_in_place_do_not_alias_the_flat_buffer(emu, two_devices):
    """AccumulateGrad keeps the views of the persistent flat gradient buffer as p.grad; a second backward without
    zero_grad(set_to_none=True) used to overwrite them in place and then add the buffer to itself (2 g2 instead of g1 + g2; exactly
    2x after zero_grad(set_to_none=False))."""
    dp, optim = _tiny_dp("sgd")
    (xa, ta), (xb, tb) = _tiny_batch(4, 1), _tiny_batch(4, 2)
    ra, rb = _tiny_reference(xa, ta), _tiny_reference(xb, tb)
    optim.zero_grad()
    _tiny_loss(dp, xa, ta).backward()
    assert len(dp._replicas) == 1
    ga = _grads(dp)
    for g, r in zip(ga, ra):
        assert torch.allclose(g, r, rtol=1e-5, atol=1e-6)
    base = dp.module._dream_flat["grads"].data_ptr()
    assert all(base <= p.grad.data_ptr() < base + 4 * dp.module._dream_flat["grads"].numel() for p in dp.parameters())
    # (1) zero_grad(set_to_none=False): p.grad stays a (zeroed) view of the buffer the next backward writes
    optim.zero_grad(set_to_none=False)
    _tiny_loss(dp, xa, ta).backward()
    for g, g0 in zip(_grads(dp), ga):
        assert torch.equal(g, g0)
    # (2) accumulation: a second backward, no zero_grad in between
    _tiny_loss(dp, xb, tb).backward()
    for g, r1, r2 in zip(_grads(dp), ra, rb):
        assert torch.allclose(g, r1 + r2, rtol=1e-5, atol=1e-6)
    # the optimizer still sees one contiguous gradient buffer, but it is not the all-reduced one: the replicas must not replay
    # the step on their own copy (which holds g_b only) -- they are refreshed from the master instead
    steps, copies = dp.stats["replica_steps"], dp.stats["param_copies"]
    optim.step()
    assert dp.stats["replica_steps"] == steps
    optim.zero_grad()
    _tiny_loss(dp, xa, ta).backward()
    assert dp.stats["param_copies"] == copies + 1
    assert torch.equal(dp._replicas[0]._dream_flat["params"], dp.module._dream_flat["params"])
    # a plain step afterwards is replayed on the replicas again
    optim.step()
    assert dp.stats["replica_steps"] == steps + 1
    assert torch.equal(dp._replicas[0]._dream_flat["params"], dp.module._dream_flat["params"])


def test_gradient_edits_between_backward_and_step_reach_the_replicas(emu, two_devices):
    """clip_grad_norm_ (or any in-place edit of p.grad) happens on the master's buffer only: step_replicas must not replay the
    un-clipped update on the replicas and stamp them as in sync."""
    dp, optim = _tiny_dp("sgd", lr=0.1)
    x, t = _tiny_batch(4, 3)
    optim.zero_grad()
    _tiny_loss(dp, x, t).backward()
    total = torch.nn.utils.clip_grad_norm_(list(dp.parameters()), max_norm=1e-3)
    assert float(total) > 1e-3                                         # the clip really scales
    steps = dp.stats["replica_steps"]
    optim.step()
    assert dp.stats["replica_steps"] == steps                          # not replayed ...
    assert not torch.equal(dp._replicas[0]._dream_flat["params"], dp.module._dream_flat["params"])
    copies = dp.stats["param_copies"]
    optim.zero_grad()
    _tiny_loss(dp, x, t).backward()                                    # ... and repaired by the flat copy of the next forward
    assert dp.stats["param_copies"] == copies + 1
    assert torch.equal(dp._replicas[0]._dream_flat["params"], dp.module._dream_flat["params"])
    optim.step()                                                       # an untouched step is replayed again
    assert dp.stats["replica_steps"] == steps + 1
    assert torch.equal(dp._replicas[0]._dream_flat["params"], dp.module._dream_flat["params"])


def test_adam_load_state_dict_drops_the_replicas_moments(emu, two_devices):
    dp, optim = _tiny_dp("adam", lr=1e-2)
    x, t = _tiny_batch(4, 4)
    for _ in range(2):
        optim.zero_grad()
        _tiny_loss(dp, x, t).backward()
        optim.step()
    assert dp._opt_state and dp.stats["replica_steps"] == 2            # the replica's own moment buffers
    saved = copy.deepcopy(optim.state_dict())
    optim.zero_grad()
    _tiny_loss(dp, x, t).backward()
    optim.step()
    optim.load_state_dict(saved)                                       # back to the moments after step 2
    assert not dp._opt_state                                           # the replica's belonged to the state just replaced
    optim.zero_grad()
    _tiny_loss(dp, x, t).backward()
    optim.step()
    assert torch.equal(dp._replicas[0]._dream_flat["params"], dp.module._dream_flat["params"])
    assert torch.equal(dp._opt_state[1][0], optim._plan["moments"][0])


def test_eight_devices_uneven_chunks(emu, monkeypatch):
    """The shape of the first 8-GPU run, on eight emulated devices.  resnet_h, 12 frames: ``Tensor.chunk`` gives six chunks of 2 --
    the last two devices idle, as nn.DataParallel.scatter -- and only the five replicas that get a chunk are built; inference
    equals the single-device result bit for bit and in order (BatchNorm folded per replica).  A training step of 12 frames over
    8 devices (six replicas) and one of 8 over 8 on the executor's bookkeeping module: ONE all-reduce per step over the
    participating flat buffers, the gradient equals the whole-batch gradient, replicas left identical.  (Eight emulated
    ResNet-101 training steps take minutes; ResNet's per-replica BatchNorm semantics in training are the two-replica tests' subject.)"""
    monkeypatch.setenv("DREAM_DP_EMULATED_DEVICES", "8")
    wts = om.recipe_weights(om.build_model("resnet_h", 7).state_dict())
    net = pc.build_network("resnet_h", "cpu", weights=wts, in_res=(32, 32))
    net.enable_evaluation()
    x12 = torch.from_numpy(cases.image_batch(12, 32, 32, seed=21))
    with torch.no_grad():
        maps, kps = net.inference(x12)
    dp = net.model
    assert len(dp.devices()) == 8 and len(x12.chunk(8)) == 6 and len(dp._replicas) == 5
    os.environ["DREAM_DP_EMULATED_DEVICES"] = "0"
    single = pc.build_network("resnet_h", "cpu", weights=wts, in_res=(32, 32))
    single.enable_evaluation()
    with torch.no_grad():
        maps1, kps1 = single.inference(x12)
    assert torch.equal(maps, maps1) and torch.equal(kps, kps1) and kps.shape == (12, 7, 2)
    os.environ["DREAM_DP_EMULATED_DEVICES"] = "8"
    reduced = []
    real = ops.allreduce_sum_
    monkeypatch.setattr(ops, "allreduce_sum_", lambda flats: (reduced.append(len(flats)), real(flats))[1])
    tdp, optim = _tiny_dp("adam")
    for n, nrep in ((12, 6), (8, 8)):
        x, t = _tiny_batch(n, 30 + n)
        optim.zero_grad()
        _tiny_loss(tdp, x, t).backward()
        for g, r in zip(_grads(tdp), _tiny_reference_at(tdp, x, t)):
            assert torch.allclose(g, r, rtol=1e-5, atol=1e-6)
        optim.step()
        assert reduced[-1] == nrep
        for rep in tdp._replicas[:nrep - 1]:
            assert torch.equal(rep._dream_flat["params"], tdp.module._dream_flat["params"])
    assert reduced == [6, 8] and len(tdp._replicas) == 7


def test_one_device_step_through_the_replica_path(emu):
    """``single_device_graphs`` (DreamNetwork.hip_graph_train): a one-device step takes the replica path -- flat
    gradient buffer, no all-reduce, no replica update -- so that it can be replayed as hipGraphs on a GPU.  Here on CPU tensors
    (nothing to capture): gradients and Adam steps equal the direct path's, accumulation over two backwards included."""
    a, oa = _tiny_dp("adam")
    b, ob = _tiny_dp("adam")
    a.single_device_graphs = True
    assert len(a.devices()) == 1 and not a.use_graphs()
    for step in range(3):
        x, t = _tiny_batch(6, 50 + step)
        for dp, optim in ((a, oa), (b, ob)):
            optim.zero_grad()
            _tiny_loss(dp, x, t).backward()
            if step == 2:                                  # a second backward before the step
                _tiny_loss(dp, x, t).backward()
        base = a.module._dream_flat["grads"]
        if step < 2:
            assert all(base.data_ptr() <= p.grad.data_ptr() < base.data_ptr() + 4 * base.numel() for p in a.parameters())
        for ga, gb in zip(_grads(a), _grads(b)):
            assert torch.allclose(ga, gb, rtol=1e-6, atol=1e-7)
        oa.step()
        ob.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)
    assert not a._replicas and a.stats["replica_steps"] == 0
    with torch.no_grad():                                  # evaluation and no-grad forwards stay on the direct path
        assert torch.equal(a(x)[0], a.module(x)[0])


def _tiny_reference_at(dp, x, t):
    """Whole-batch gradient of the tiny module at the data-parallel master's current parameters (plain autograd)."""
    ref = _TinyReplica()
    with torch.no_grad():
        ref.w.copy_(dp.module.w)
        ref.b.copy_(dp.module.b)
    ((ref(x)[0] - t) ** 2).mean().backward()
    return [ref.w.grad, ref.b.grad]
