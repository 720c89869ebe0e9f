"""Single-process data parallelism behind ``training.platform.gpu_ids`` -- what ``torch.nn.DataParallel(model,
device_ids=gpu_ids).cuda()`` does for the reference (/root/reference/dream/network.py:185,244-256,281-284; ``-g`` of
/root/reference/scripts/train_network.py:755-762: "Nothing specified means all GPUs"), re-designed for a node of MI355Xs:

  * PERSISTENT replicas.  nn.DataParallel re-broadcasts every parameter to every device on every forward (88.9 - 220 MB x 7
    peers) and re-creates the replica modules each call.  Here every listed device keeps its own copy of the
    model; all parameters of a copy are views into ONE flat buffer, so a refresh after an optimizer step or a
    ``load_state_dict`` is one device-to-device copy per replica over xGMI -- and it only happens when a parameter's version
    counter moved.
  * The batch is split into contiguous chunks of dim 0 (``Tensor.chunk``, as ``DataParallel.scatter`` does); every chunk
    runs forward (and backward) on its device from its own host thread; belief maps are concatenated on ``device_ids[0]``
    in the original order, keypoints of ``DreamNetwork.inference`` are extracted on each device and concatenated on the host.
  * Training: one autograd node for the whole data-parallel network.  Its backward scatters dL/d(maps), runs every replica's
    backward plan, packs each replica's parameter gradients into the flat layout, moves the flat buffers to ``device_ids[0]``
    (peer copies, one per replica, each on its own xGMI link), sums them there and hands views of the sum to autograd, so the
    optimizer sees one contiguous gradient buffer (a single Adam launch, dream_amd/optim.py).  BatchNorm: per-replica batch
    statistics, running statistics kept by replica 0 = the module whose ``state_dict()`` callers save -- as DataParallel.
  * Under torchrun (LOCAL_RANK set, one process per GPU -- the path bench.py uses for N > 1) the wrapper is a pass-through and
    the gradient exchange is the bucketed RCCL all-reduce inside the model's own autograd node (dream_amd/models.py).

The host side is Python threads: kernel launches go through ctypes, which releases the GIL for the duration of the call.
"""
import copy
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import torch
import torch.nn as nn

from . import ops

ALIGN = 64                      # floats: every tensor starts on a 256-byte boundary of the flat buffer


class FlatLayout:
    """A list of fp32 tensors re-homed as views of one contiguous buffer (same order, 256-byte aligned starts)."""

    def __init__(self, tensors):
        self.offsets, o = [], 0
        for t in tensors:
            self.offsets.append(o)
            o += (t.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = o
        self.shapes = [tuple(t.shape) for t in tensors]
        self.numels = [t.numel() for t in tensors]

    def views(self, flat):
        return [flat[o:o + n].view(s) for o, n, s in zip(self.offsets, self.numels, self.shapes)]


def flatten_tensors_(tensors, set_data):
    """Move ``tensors`` (same device, fp32) into one flat buffer; ``set_data(i, view)`` re-points tensor i at its view.
    -> (flat, layout) or (None, None) when there is nothing to flatten."""
    tensors = list(tensors)
    if not tensors:
        return None, None
    dev = tensors[0].device
    assert all(t.device == dev and t.dtype == torch.float32 for t in tensors)
    layout = FlatLayout(tensors)
    flat = torch.zeros((layout.total,), dtype=torch.float32, device=dev)
    with torch.no_grad():
        for i, (t, v) in enumerate(zip(tensors, layout.views(flat))):
            v.copy_(t)
            set_data(i, v)
    return flat, layout


def flatten_module_(module):
    """Re-home the fp32 parameters and the fp32 buffers (BatchNorm running statistics) of ``module`` in two flat buffers.
    Idempotent; must be repeated after ``module.to(...)`` (which re-allocates every tensor).  Returns the record that is also
    stored as ``module._dream_flat``: {"params": flat, "param_layout", "param_list", "buffers": flat | None, ...}."""
    params = [p for p in module.parameters() if p.dtype == torch.float32]
    bufs = [(m, n) for m in module.modules() for n, b in m._buffers.items() if b is not None and b.dtype == torch.float32]

    def set_param(i, v):
        params[i].data = v

    def set_buf(i, v):
        m, n = bufs[i]
        m._buffers[n] = v

    pflat, playout = flatten_tensors_([p.data for p in params], set_param)
    bflat, blayout = flatten_tensors_([m._buffers[n] for m, n in bufs], set_buf)
    rec = {"params": pflat, "param_layout": playout, "param_list": params, "buffers": bflat, "buffer_layout": blayout,
           "buffer_list": bufs}
    object.__setattr__(module, "_dream_flat", rec)
    return rec


def flat_is_intact(module):
    """Do the parameters still live where flatten_module_ put them?  (``.to()``, ``.float()``, ``p.data = ...`` move them.)"""
    rec = getattr(module, "_dream_flat", None)
    if rec is None or rec["params"] is None:
        return False
    base, lay = rec["params"].data_ptr(), rec["param_layout"]
    return all(p.data_ptr() == base + 4 * o for p, o in zip(rec["param_list"], lay.offsets))


def _version_stamp(module):
    rec = module._dream_flat
    return (sum(p._version for p in rec["param_list"]),
            sum(m._buffers[n]._version for m, n in rec["buffer_list"]),
            rec["params"].data_ptr())


class _DataParallelFunction(torch.autograd.Function):
    """The whole data-parallel network as one autograd node: inputs = the batch and the master's parameters."""

    @staticmethod
    def forward(ctx, dp, x, *params):
        outs, ctxs, sizes = dp._forward_shards(x.detach(), save=True)
        ctx.dp, ctx.ctxs, ctx.sizes = dp, ctxs, sizes
        return tuple(dp._gather(outs))

    @staticmethod
    def backward(ctx, *grad_outs):
        grads = ctx.dp._backward_shards(ctx.ctxs, grad_outs, ctx.sizes)
        ctx.ctxs = None
        return (None, None) + tuple(grads)


class DreamDataParallel(nn.Module):
    """Drop-in for ``torch.nn.DataParallel`` on the path of dream/network.py:244-256: ``.module``, ``module.``-prefixed
    ``state_dict()`` keys, ``device_ids`` (None / empty = every visible device), input on ``device_ids[0]``, outputs gathered
    on ``device_ids[0]``."""

    def __init__(self, module, device_ids=None):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids else None
        object.__setattr__(self, "_replicas", [])         # modules for devices[1:] -- deliberately not registered
        object.__setattr__(self, "_devices", None)
        object.__setattr__(self, "_pool", None)
        object.__setattr__(self, "_stamp", None)
        object.__setattr__(self, "_lock", threading.Lock())

    # ---- devices ----------------------------------------------------------------------------------------------------------
    def devices(self):
        """Devices the next call is spread over (resolved once, after the module has been moved to its device)."""
        if self._devices is None:
            master = next(self.module.parameters()).device
            n_emulated = int(os.environ.get("DREAM_DP_EMULATED_DEVICES", "0"))     # test hook: replicas on CPU tensors
            if master.type != "cuda":
                devs = [master] * max(1, n_emulated)
            elif "LOCAL_RANK" in os.environ or _distributed_world() > 1:
                devs = [master]                               # one process per GPU: parallelism is across processes
            else:
                ids = self.device_ids if self.device_ids else list(range(torch.cuda.device_count()))
                devs = [torch.device("cuda", int(i)) for i in ids]
                if devs[0] != master:
                    raise RuntimeError("module must have its parameters and buffers on device_ids[0] (%s) but found one of "
                                       "them on device: %s" % (devs[0], master))       # nn.DataParallel's own message
            object.__setattr__(self, "_devices", devs)
        return self._devices

    def n_devices(self, batch):
        return max(1, min(len(self.devices()), int(batch)))

    # ---- replicas ---------------------------------------------------------------------------------------------------------
    def flatten_parameters(self):
        """Called by DreamNetwork once the model sits on its device: parameters become views of one flat buffer."""
        return flatten_module_(self.module)

    def _replica(self, i):
        return self.module if i == 0 else self._replicas[i - 1]

    def _ensure_replicas(self, n):
        if not flat_is_intact(self.module):
            flatten_module_(self.module)
            object.__setattr__(self, "_stamp", None)
        devs = self.devices()
        while len(self._replicas) < n - 1:
            dev = devs[len(self._replicas) + 1]
            rep = copy.deepcopy(self.module)
            if dev.type == "cuda":
                with torch.cuda.device(dev):
                    rep = rep.to(dev)
            flatten_module_(rep)
            for prm in rep.parameters():
                prm.requires_grad_(False)
            self._replicas.append(rep)
            object.__setattr__(self, "_stamp", None)
        if self._pool is None and n > 1 and devs[0].type == "cuda":
            object.__setattr__(self, "_pool", ThreadPoolExecutor(max_workers=len(devs) - 1, thread_name_prefix="dream-dp"))

    def _sync_replicas(self, n):
        """Refresh the replicas from the master when a parameter / buffer changed: one flat copy each."""
        stamp = _version_stamp(self.module)
        for rep in self._replicas[:n - 1]:
            rep.train(self.module.training)
            for attr in ("precision", "conv_algorithm"):
                if hasattr(self.module, attr) and getattr(rep, attr) != getattr(self.module, attr):
                    setattr(rep, attr, getattr(self.module, attr))
        if stamp == self._stamp:
            return
        src = self.module._dream_flat
        with torch.no_grad():
            for rep in self._replicas:
                dst = rep._dream_flat
                dst["params"].copy_(src["params"], non_blocking=True)
                for prm in dst["param_list"]:
                    ops.bump_version(prm)
                if src["buffers"] is not None:
                    dst["buffers"].copy_(src["buffers"], non_blocking=True)
                    for m, name in dst["buffer_list"]:
                        ops.bump_version(m._buffers[name])
        object.__setattr__(self, "_stamp", stamp)

    def train(self, mode=True):
        super().train(mode)
        for rep in self._replicas:
            rep.train(mode)
        return self

    # ---- execution --------------------------------------------------------------------------------------------------------
    def _run(self, jobs):
        """jobs[i]() runs on device i; job 0 in the calling thread, the others in the pool (serially when emulated)."""
        devs = self.devices()

        def on_device(i):
            if devs[i].type != "cuda":
                return jobs[i]()
            with torch.cuda.device(devs[i]):
                return jobs[i]()
        if self._pool is None:
            return [on_device(i) for i in range(len(jobs))]
        futures = [self._pool.submit(on_device, i) for i in range(1, len(jobs))]
        first = on_device(0)
        return [first] + [f.result() for f in futures]

    def _scatter(self, x, n):
        devs = self.devices()
        chunks = x.chunk(n, dim=0)
        return [c.contiguous() if d == x.device else c.to(d, non_blocking=True) for c, d in zip(chunks, devs)]

    def _forward_shards(self, x, save, post=None):
        """-> (outs[i] = list of output tensors of replica i (+ post(outs) appended when given), ctxs, chunk sizes)."""
        with self._lock:
            n = self.n_devices(x.shape[0])
            self._ensure_replicas(n)
            self._sync_replicas(n)
            xs = self._scatter(x, n)

            def job(i):
                def run():
                    with torch.no_grad():
                        outs, c = self._replica(i).dp_forward(xs[i], save)
                        return (outs + [post(outs)] if post is not None else outs), c
                return run
            res = self._run([job(i) for i in range(len(xs))])
        return [r[0] for r in res], [r[1] for r in res], [int(c.shape[0]) for c in xs]

    def _gather(self, outs, upto=None):
        master = self.devices()[0]
        k = len(outs[0]) if upto is None else upto
        return [torch.cat([o[j] if o[j].device == master else o[j].to(master, non_blocking=True) for o in outs], dim=0)
                for j in range(k)]

    def _backward_shards(self, ctxs, grad_outs, sizes):
        devs = self.devices()
        n = len(sizes)
        master_rec = self.module._dream_flat
        params = self.module.dp_parameters()
        base = master_rec["params"].data_ptr()
        offsets = [(prm.data_ptr() - base) // 4 for prm in params]          # the gradient buffer mirrors the parameter buffer
        numels = [prm.numel() for prm in params]
        total = master_rec["param_layout"].total
        splits = [g.split(sizes, dim=0) if g is not None else [None] * n for g in grad_outs]

        def job(i):
            def run():
                with torch.no_grad():
                    gos = [None if s[i] is None else (s[i].contiguous() if s[i].device == devs[i] else s[i].to(devs[i], non_blocking=True))
                           for s in splits]
                    grads = self._replica(i).dp_backward(ctxs[i], gos)
                    flat = torch.zeros((total,), dtype=torch.float32, device=devs[i])
                    views = [flat[o:o + m].view_as(g) for o, m, g in zip(offsets, numels, grads)]
                    torch._foreach_copy_(views, [g.contiguous() for g in grads])
                    return flat
            return run
        flats = self._run([job(i) for i in range(n)])
        total_flat = flats[0]
        for f in flats[1:]:                                                  # peer copy + add on device_ids[0]
            ops.add_(total_flat, f if f.device == total_flat.device else f.to(total_flat.device, non_blocking=True))
        return [total_flat[o:o + m].view(prm.shape) for o, m, prm in zip(offsets, numels, params)]

    # ---- nn.Module interface ------------------------------------------------------------------------------------------------
    def forward(self, x, *args, **kwargs):
        if self.n_devices(x.shape[0]) == 1 or args or kwargs:
            return self.module(x, *args, **kwargs)
        params = self.module.dp_parameters()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params) and self.module.dp_trainable():
            outs = list(_DataParallelFunction.apply(self, x, *params))
        else:
            outs, _, _ = self._forward_shards(x, save=False)
            outs = self._gather(outs)
        return self.module.dp_finish(outs)

    def inference_shards(self, x, post):
        """No-grad forward of every chunk on its device followed by ``post(outputs)`` on the same device (the peak
        extraction of DreamNetwork.inference).  -> (outputs gathered on device_ids[0], [post result of each chunk])."""
        outs, _, _ = self._forward_shards(x, save=False, post=post)
        return self._gather(outs, upto=len(outs[0]) - 1), [o[-1] for o in outs]


def _distributed_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
