"""CPU, world_size 2, gloo: the N>1 path of the hot loop.  Each rank runs
DreamNetwork.train() on its half of the batch (kernels under the SIMT emulator); the flat gradient buffer
is all-reduced inside the network's autograd node.  Afterwards every rank must hold identical parameters,
equal (to fp32 summation-order tolerance) to a single-process step on the full batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup_paths():
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _train_once(x, t, steps=1):
    import cases
    import parity_checks as pc
    from oracle import models as om
    w = om.recipe_weights(om.build_model("vgg_q", 7).state_dict(), cases.TRAIN_FINAL_KEYS, cases.TRAIN_FINAL_SCALE)
    net = pc.build_network("vgg_q", "cpu", weights=w, optimizer="sgd", lr=1e-6, in_res=(32, 32))
    net.enable_training()
    losses = [net.train([x], t).item() for _ in range(steps)]
    return losses, {k: v.detach().clone() for k, v in net.model.named_parameters()}


def _worker(rank, world, port, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
    import cases
    from emu_util import emulated_hip
    x = torch.from_numpy(cases.image_batch(2, 32, 32, seed=9))
    t = torch.from_numpy(cases.target_batch(2, 7, (8, 8), in_wh=(32, 32), seed=9))
    from dream_amd import models
    models._OverlappedAllReduce.BUCKET_BYTES = 1 << 20      # several asynchronous buckets in flight during backward
    with emulated_hip():
        losses, params = _train_once(x[rank:rank + 1], t[rank:rank + 1])
    torch.save({"losses": losses, "params": params}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gradients_two_ranks(tmp_path):
    _setup_paths()
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    build_emu.build()                      # build the emulated kernels once, before the two ranks need them
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k          # replicas stay bit-identical
    import cases
    from emu_util import emulated_hip
    x = torch.from_numpy(cases.image_batch(2, 32, 32, seed=9))
    t = torch.from_numpy(cases.target_batch(2, 7, (8, 8), in_wh=(32, 32), seed=9))
    with emulated_hip():
        losses, params = _train_once(x, t)
    # global mean loss = mean of the two local mean losses (equal chunks)
    assert np.allclose(np.mean([r0["losses"], r1["losses"]], axis=0), losses, rtol=1e-5)
    for k in params:
        upd = (params[k] - r0["params"][k]).abs().max().item()
        assert upd <= 1e-6 + 1e-4 * params[k].abs().max().item(), (k, upd)


def test_allreduce_is_noop_single_process():
    _setup_paths()
    from dream_amd.models import allreduce_gradients
    g = [torch.randn(3, 4), torch.randn(5)]
    out = allreduce_gradients(g)
    assert all(a is b for a, b in zip(g, out))
