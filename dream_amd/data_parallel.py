"""Single-process data parallelism behind ``training.platform.gpu_ids`` -- what ``torch.nn.DataParallel(model,
device_ids=gpu_ids).cuda()`` does for the reference (/root/reference/dream/network.py:185,244-256,281-284; ``-g`` of
/root/reference/scripts/train_network.py:755-762: "Nothing specified means all GPUs"), re-designed for a node of MI355Xs:

  * PERSISTENT replicas.  nn.DataParallel re-broadcasts every parameter to every device on every forward (88.9 - 220 MB x 7
    peers) and re-creates the replica modules each call.  Here every listed device keeps its own copy of the
    model; all parameters of a copy are views into ONE flat buffer.
  * The batch is split into contiguous chunks of dim 0 (``Tensor.chunk``, as ``DataParallel.scatter`` does); every chunk
    runs forward (and backward) on its device from its own host thread; belief maps are concatenated on ``device_ids[0]``
    in the original order, keypoints of ``DreamNetwork.inference`` are extracted on each device and concatenated on the host.
  * Training: one autograd node for the whole data-parallel network.  Its backward scatters dL/d(maps), runs every replica's
    backward plan, packs each replica's parameter gradients into that replica's flat gradient buffer (same layout as the flat
    parameter buffer) and sums the buffers in place with ONE RCCL all-reduce over xGMI (ops.allreduce_sum_ ->
    csrc/collective.hip: ncclCommInitAll group of the listed devices).  Autograd receives views of the master's (summed)
    buffer, so the optimizer sees one contiguous gradient buffer: a single Adam launch on the master -- and, through
    ``step_replicas``, the identical single launch on every replica with that replica's own copy of the summed gradients and
    its own moment buffers.  No gather on GPU 0, no parameter copy after the step: the replicas stay identical because
    they apply the same update to the same numbers.  (An optimizer that is not dream_amd's leaves the replicas stale; the
    version stamp notices and they are refreshed by one flat peer copy each.)  BatchNorm: per-replica batch statistics,
    running statistics kept by replica 0 = the module whose ``state_dict()`` callers save -- as DataParallel.
  * The host is off the critical path: a replica's launch sequence for one input shape -- ~1 700 launches for a ResNet-101
    training step, each a Python ``torch.empty`` + ctypes call under the GIL, 30 ms of host time per replica-step -- is
    captured into hipGraphs the second time the shape is seen (forward graph, backward graph, one memory pool) and replayed
    from then on: per replica and step the GIL is held for two graph launches and the input copies.  Weight packing runs inside
    the training graphs (weights change every step), evaluation graphs are keyed on the parameter versions.
  * Under torchrun (LOCAL_RANK set, one process per GPU -- the path bench.py uses for N > 1) the wrapper is a pass-through and
    the gradient exchange is the bucketed RCCL all-reduce inside the model's own autograd node (dream_amd/models.py).
"""
import copy
import gc
import os
import threading
import weakref
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


def reset_weight_caches(module):
    """Drop every packed / folded copy of the weights a model keeps for the kernels (they are rebuilt on first use)."""
    from . import models
    for m in module.modules():
        if "_packed" in m.__dict__:
            m._packed = models._PackedCache()
        for name in ("_cache", "_aux"):
            if name in m.__dict__:
                setattr(m, name, {})
        if m.__dict__.get("_pack_pending") is not None:
            m._pack_join()                                  # a re-pack on the second stream still writes into the old copies
        for name in ("_pack_state", "_pack_records", "_pack_pending"):       # ResnetSimple._repack_weights: its graph writes into the old copies
            m.__dict__.pop(name, None)


def _param_stamp(module):
    rec = module._dream_flat
    return (sum(p._version for p in rec["param_list"]), rec["params"].data_ptr())


def _buffer_stamp(module):
    rec = module._dream_flat
    return sum(m._buffers[n]._version for m, n in rec["buffer_list"])


def _settings(module):
    return tuple(getattr(module, a, None) for a in ("precision", "conv_algorithm", "conv1x1_algorithm", "convT_algorithm"))


class _Busy:
    """Token of a captured training forward whose backward has not run yet (the saved activations live in the graph's pool:
    a second forward of the same graph would overwrite them).  Released by the backward or when autograd drops the context."""

    def __init__(self, entry):
        self.entry = entry
        entry["busy"] = True
        weakref.finalize(self, _Busy._release, entry)

    @staticmethod
    def _release(entry):
        entry["busy"] = False


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


def _probe_sync(where):
    """DREAM_DP_PROBE_SYNC=<where> (tools/dp_exchange_probe.py): a device-wide synchronisation at one point of the data-parallel step --
    backward_begin | backward_end | exchange_begin | exchange_mid | exchange_end.  A step's results must not depend on it; they did, until
    round 6 replaced the memset nodes of the captured graphs by kernels (csrc/common.h: dream_zero_words)."""
    if os.environ.get("DREAM_DP_PROBE_SYNC", "") == where and torch.cuda.is_available():
        torch.cuda.synchronize()


class _EarlyBucket:
    """One replica's early bucket of the single-process gradient exchange (round 6; dream/network.py:244-256,335: nn.DataParallel
    reduces the replicas' gradients after the backward pass).  ``dp_parameters()[k:]`` -- for ResNet-101 everything from layer3 up,
    97 % of the 216 MB -- is final well before the backward pass ends; the replica's backward plan packs it into its flat gradient
    buffer by a leaf on the second stream and records ``event`` behind that leaf (models._early_bucket_hook); the exchange of
    ``gflat[lo:]`` then waits for the event only and runs on the device's exchange stream beside the rest of the backward pass.
    The object is handed to ``dp_backward`` in the reducer slot (``add`` is what the gradient containers call per assignment)."""

    def __init__(self, k, marker, views, lo):
        self.early_k, self.early_marker, self.views, self.lo = k, marker, views, lo      # views: gflat views of dp_parameters()[k:]
        self.event = None
        self.packed = False            # the plan contains the pack (set when the plan runs or is captured)
        self.marked = False            # the event was recorded in the step that is running

    def add(self, g):
        pass

    def pack_early(self, grads):
        torch._foreach_copy_(self.views, [g.contiguous() for g in grads])
        self.packed = True

    def mark_early(self, stream):
        if stream is not None:
            if self.event is None:
                self.event = torch.cuda.Event()
            self.event.record(stream)
        self.marked = True


class _SplitCapture:
    """A replica's backward captured as a SEQUENCE of hipGraphs (``DREAM_TRAIN_GRAPH_SPLIT=n``: n weight-gradient leaves per
    segment) instead of one graph with a forked branch -- see models._DeferredSide for why.  ``plan`` = [("main", graph) |
    ("side", graph) | ("join",)] in capture order.  Main segments share the replica's pool (with the forward graph: they replay in
    capture order on one stream); the leaf segments have a pool of their own, because they replay CONCURRENTLY with the main
    segments that follow them -- a block freed inside a leaf segment must not be handed to a later main segment.  What crosses over
    is kept alive by construction: leaf inputs (main pool) until join(), leaf outputs = the weight gradients (side pool) until
    the last main segment has packed them."""

    def __init__(self, pool, leaves, device):
        self.pool, self.side_pool, self.leaves = pool, None, max(1, int(leaves))
        self.plan, self.cur = [], None
        from . import models
        # The leaf segments are replayed to the device's LIVE second stream (the one eager steps use: a stream that is known to run
        # beside the main stream) and captured on the process's capture stream for leaves (captures are taken one at a time,
        # process-wide: _capture_lock) -- never on the live stream itself: with replicas that share a device (gpu_ids=[0, 0, 0, 0], the
        # rehearsal on a one-GPU box) another replica's thread may be replaying to it at that moment (round-5 advice).  Both are
        # streams of the process's own (_hip.own_stream), not drawn from torch's pool of 32, where a long-running process is handed
        # the same stream twice: a capture begun on such an alias of the live stream failed another replica's replay with
        # "Cannot prepare for replay during capturing stage" (tools/dp_exchange_probe.py, sixth run in one process).
        from . import _hip
        self.side = models._SideStream.live_stream(device)               # replays the leaf segments
        self.capture_side = _hip.own_stream(device, "capture-leaves")    # captures them
        self.mark = torch.zeros(1, device=device)               # one tiny node per main segment: never an empty graph

    def _begin(self):
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def _end(self):
        self.mark.add_(1.0)
        self.cur.capture_end()
        self.plan.append(("main", self.cur))
        self.cur = None

    def cut(self, fns, join=False, after=None):
        """Called by models._DeferredSide from inside the backward: close the running main segment, capture ``fns`` (the collected
        leaves) on the second stream, open the next main segment; ``join``: the next main segment waits for every leaf segment;
        ``after``: called with the leaf stream at every replay once the leaf segment has been launched (and once now)."""
        self._end()
        if fns:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(self.capture_side):
                if self.side_pool is None:
                    graph.capture_begin(capture_error_mode="thread_local")
                else:
                    graph.capture_begin(pool=self.side_pool, capture_error_mode="thread_local")
                try:
                    with ops.wgrad_width(ops.SIDE_WGRAD_WIDTH):       # leaves beside the chain: as the eager step's second stream
                        for fn in fns:
                            fn()
                finally:
                    graph.capture_end()
            if self.side_pool is None:
                self.side_pool = graph.pool()
            self.plan.append(("side", graph))
        if after is not None:
            self.plan.append(("call", after))
            after(None)                                         # (capture time: nothing runs, nothing to record -- the plan's state only)
        if join:
            self.plan.append(("join",))
        self._begin()

    def capture(self, fn):
        from . import models
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        from . import _hip
        stream = _hip.own_stream(self.mark.device, "capture")   # captures cannot be taken on the default stream
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            models._split_capture.ctl = self
            try:
                self._begin()
                try:
                    fn()
                except BaseException:
                    if self.cur is not None:                    # leave no capture open behind the error that is being reported
                        try:
                            self.cur.capture_end()
                        except Exception:
                            pass
                    raise
                self._end()
            finally:
                models._split_capture.ctl = None
        torch.cuda.current_stream().wait_stream(stream)

    def replay(self):
        main, behind = torch.cuda.current_stream(), False
        for op in self.plan:
            if op[0] == "main":
                op[1].replay()
            elif op[0] == "side":
                self.side.wait_stream(main)                     # the leaves read what the segments so far have produced
                with torch.cuda.stream(self.side):
                    op[1].replay()
                behind = True
            elif op[0] == "call":
                op[1](self.side)
            else:
                main.wait_stream(self.side)
                behind = False
        if behind:
            main.wait_stream(self.side)


class DreamDataParallel(nn.Module):
    """Drop-in for ``torch.nn.DataParallel`` on the path of dream/network.py:244-256: ``.module``, ``module.``-prefixed
    ``state_dict()`` keys, ``device_ids`` (None / empty = every visible device), input on ``device_ids[0]``, outputs gathered
    on ``device_ids[0]``."""

    _capture_lock = threading.Lock()              # hipGraph captures are taken one at a time, process-wide

    def __init__(self, module, device_ids=None):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids else None
        object.__setattr__(self, "_replicas", [])         # modules for devices[1:] -- deliberately not registered
        object.__setattr__(self, "_devices", None)
        object.__setattr__(self, "_pool", None)
        object.__setattr__(self, "_pstamp", None)         # parameter / buffer versions of the master the replicas agree with
        object.__setattr__(self, "_bstamp", None)
        object.__setattr__(self, "_lock", threading.Lock())
        object.__setattr__(self, "_graphs", {})           # (replica index, key) -> captured launch sequence
        object.__setattr__(self, "_reduced", 0)           # replicas whose flat gradient buffer holds this step's all-reduced sum
        object.__setattr__(self, "_opt_state", {})        # replica index -> optimizer state buffers on that replica's device
        object.__setattr__(self, "_xstreams", {})         # device index -> the stream the early bucket of the gradient exchange runs on
        object.__setattr__(self, "_tick", [0])            # use counter for the LRU of captured graphs
        object.__setattr__(self, "_grad_version", None)   # version of the master's flat gradient buffer right after the all-reduce
        # Opt-in (DreamNetwork.hip_graph_train / DREAM_TRAIN_GRAPH=1): a training step on ONE device also runs as hipGraph replays
        # (forward, backward) instead of ~2000 launches from Python -- what every replica of a multi-device step already does.  A
        # ResNet-101 step at 16 frames leaves the GPU idle 11 % of the time waiting for the host between its small kernels
        # (profiles/r04_bench_resnet_h_train16_concurrency.txt).  Same kernels in the same order: bit-identical.
        self.single_device_graphs = os.environ.get("DREAM_TRAIN_GRAPH", "0") == "1"
        # DREAM_TRAIN_GRAPH_SPLIT=n: a captured backward becomes a sequence of graphs, n weight-gradient leaves per segment, the leaf
        # segments replayed on a live second stream (_SplitCapture); 0: one graph, the leaves a forked branch inside it.  Unset (None):
        # 12, measured on the one-device step (resnet_h, 16 frames: 342 -> 363-365 frames/s, eager 367; 8 leaves: 360-364, 5: 359.5;
        # vgg_q at 16 frames 585 -> 590, eager 586.5; profiles/r05_ab_train_graph.txt).  Round 6: also the default of the replicas of a
        # multi-device step -- each replica is its device's only user, i.e. exactly the one-device step that was measured (four
        # replicas on one GPU pass their bit-for-bit test with it; profiles/r06_rehearsal_*).
        env = os.environ.get("DREAM_TRAIN_GRAPH_SPLIT")
        self.graph_split_leaves = int(env) if env not in (None, "") else None
        # statistics of the last step (tests, bench): hipGraph replays / eager replica runs / captures
        object.__setattr__(self, "stats", {"replays": 0, "eager": 0, "captures": 0, "param_copies": 0, "replica_steps": 0})

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
        """Devices a batch is spread over = the number of chunks ``Tensor.chunk`` makes of it (12 frames over 8 devices: six
        chunks of 2, the last two devices idle -- as nn.DataParallel.scatter)."""
        n = max(1, min(len(self.devices()), int(batch)))
        per = -(-int(batch) // n)
        return max(1, -(-int(batch) // per))

    def use_graphs(self):
        """Replica launch sequences as hipGraphs: real GPUs, more than one replica (or ``single_device_graphs``), not switched
        off (DREAM_DP_GRAPHS=0)."""
        devs = self.devices()
        on = ((len(devs) > 1 or self.single_device_graphs) and devs[0].type == "cuda"
              and os.environ.get("DREAM_DP_GRAPHS", "1") != "0")
        if on:
            from . import models
            models._SideStream.forbid_low_priority()      # see there: low-priority streams and replayed graphs do not mix
        return on

    # ---- replicas ---------------------------------------------------------------------------------------------------------
    def flatten_parameters(self):
        """Called by DreamNetwork once the model sits on its device: parameters become views of one flat buffer."""
        return flatten_module_(self.module)

    def _replica(self, i):
        return self.module if i == 0 else self._replicas[i - 1]

    def _ensure_replicas(self, n):
        if not flat_is_intact(self.module):
            flatten_module_(self.module)
            self._invalidate()
        devs = self.devices()
        while len(self._replicas) < n - 1:
            dev = devs[len(self._replicas) + 1]
            # the master's packed weight copies and its flat record stay with the master: a replica builds its own on its device
            memo = {id(v): None for m in self.module.modules()
                    for v in (m.__dict__.get("_packed"), m.__dict__.get("_cache"), m.__dict__.get("_aux"), m.__dict__.get("_dream_flat"),
                              m.__dict__.get("_pack_state"), m.__dict__.get("_pack_records"), m.__dict__.get("_pack_pending"))
                    if v is not None}
            rep = copy.deepcopy(self.module, memo)
            rep.__dict__.pop("_dream_flat", None)
            reset_weight_caches(rep)
            if dev.type == "cuda":
                with torch.cuda.device(dev):
                    rep = rep.to(dev)
            flatten_module_(rep)
            for prm in rep.parameters():
                prm.requires_grad_(False)
            self._replicas.append(rep)
            self._invalidate()
        if self._pool is None and n > 1 and devs[0].type == "cuda":
            object.__setattr__(self, "_pool", ThreadPoolExecutor(max_workers=len(devs) - 1, thread_name_prefix="dream-dp"))

    def _invalidate(self):
        object.__setattr__(self, "_pstamp", None)
        object.__setattr__(self, "_bstamp", None)
        self._graphs.clear()
        self._opt_state.clear()

    def _sync_replicas(self, n):
        """Bring the replicas in line with the master where they are not: parameters by one flat peer copy each (only when
        something other than dream_amd's optimizers changed them: load_state_dict, a foreign optimizer, manual edits);
        BatchNorm running statistics by one small copy each, and only for evaluation (training uses batch statistics)."""
        for rep in self._replicas[:n - 1]:
            rep.train(self.module.training)
            for attr in ("precision", "conv_algorithm", "conv1x1_algorithm", "convT_algorithm"):
                if hasattr(self.module, attr) and getattr(rep, attr) != getattr(self.module, attr):
                    setattr(rep, attr, getattr(self.module, attr))
        src = self.module._dream_flat
        pstamp = _param_stamp(self.module)
        if pstamp != self._pstamp:
            with torch.no_grad():
                for rep in self._replicas:
                    dst = rep._dream_flat
                    dst["params"].copy_(src["params"], non_blocking=True)
                    for prm in dst["param_list"]:
                        ops.bump_version(prm)
            self.stats["param_copies"] += len(self._replicas)
            self._opt_state.clear()                         # replica moments belong to the parameter history that just ended
            object.__setattr__(self, "_pstamp", pstamp)
        if src["buffers"] is not None and not self.module.training:
            bstamp = _buffer_stamp(self.module)
            if bstamp != self._bstamp:
                with torch.no_grad():
                    for rep in self._replicas:
                        dst = rep._dream_flat
                        dst["buffers"].copy_(src["buffers"], non_blocking=True)
                        for m, name in dst["buffer_list"]:
                            ops.bump_version(m._buffers[name])
                object.__setattr__(self, "_bstamp", bstamp)

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

    # ---- one replica's forward / backward, eager or as hipGraph replay ------------------------------------------------------
    def _graph_key(self, x, save, post_key):
        rep0 = self.module
        key = (tuple(x.shape), bool(save), bool(rep0.training), _settings(rep0), post_key)
        if not (rep0.training and save):
            # evaluation-type graphs read packed weights that were built outside the graph: valid for these versions only
            key += (_param_stamp(rep0), _buffer_stamp(rep0) if rep0._dream_flat["buffers"] is not None else 0)
        return key

    def _replica_forward(self, i, x, save, post, post_key):
        """-> (outputs (+ post result), context for _replica_backward)."""
        rep = self._replica(i)
        if not self.use_graphs() or (rep.training and not save):
            # (a train-mode forward without a backward updates the BatchNorm statistics it is keyed on: it would never replay)
            return self._eager_forward(rep, x, save, post)
        key = self._graph_key(x, save, post_key)
        entry = self._graphs.get((i, key))
        if entry is None:
            # evaluation graphs of older parameter versions can never be replayed again: drop them (and their memory pools)
            for k in [k for k in self._graphs if k[0] == i and k[1][:5] == key[:5] and k[1] != key]:
                del self._graphs[k]
            # every captured (replica, shape) keeps a private memory pool -- activations, packed weights, the backward pass: several
            # GB for a ResNet-101 training step -- so ragged last batches or a varying batch size must not pile them up: least
            # recently used first, beyond DREAM_DP_MAX_GRAPHS (4) per replica
            mine = sorted((k for k in self._graphs if k[0] == i and not self._graphs[k]["busy"]), key=lambda k: self._graphs[k]["tick"])
            for k in mine[:max(0, len(mine) + 1 - int(os.environ.get("DREAM_DP_MAX_GRAPHS", "4")))]:
                del self._graphs[k]
            entry = self._graphs[(i, key)] = {"seen": 0, "fwd": None, "bwd": None, "busy": False, "tick": 0}
        entry["seen"] += 1
        self._tick[0] += 1
        entry["tick"] = self._tick[0]
        if entry["fwd"] is None and (entry["seen"] < 2 or entry["busy"]):
            self.stats["eager"] += 1
            return self._eager_forward(rep, x, save, post)          # first sighting of this shape: run it as it comes
        if entry["busy"]:
            self.stats["eager"] += 1
            return self._eager_forward(rep, x, save, post)          # its saved activations still await their backward
        if entry["fwd"] is None:
            self._capture_forward(entry, rep, x, save, post)
        entry["x"].copy_(x, non_blocking=True)
        entry["fwd"].replay()
        for t in entry["bumps"]:
            ops.bump_version(t)
        self.stats["replays"] += 1
        return entry["outs"], (("graph", entry, _Busy(entry)) if save else None)

    @staticmethod
    def _eager_forward(rep, x, save, post):
        with torch.no_grad():
            outs, c = rep.dp_forward(x, save)
            return (outs + [post(outs)] if post is not None else outs), (("eager", c) if save else None)

    def _capture_forward(self, entry, rep, x, save, post):
        with DreamDataParallel._capture_lock, torch.no_grad():
            if rep.training and save:
                reset_weight_caches(rep)            # weights change every step: their packing belongs inside the graph
            entry["x"] = x.clone()
            torch.cuda.current_stream().synchronize()
            graph = torch.cuda.CUDAGraph()
            with ops.log_bumps() as bumps, torch.cuda.graph(graph, capture_error_mode="thread_local"):
                outs, c = rep.dp_forward(entry["x"], save)
                if post is not None:
                    outs = outs + [post(outs)]
            entry.update(fwd=graph, outs=outs, saved=c, bumps=list(bumps), pool=graph.pool())
            self.stats["captures"] += 1

    def _early_bucket(self, i, gflat, offsets, numels, n):
        """The early bucket of replica i for this step's exchange, or None (one device and no forced exchange, DREAM_DP_BUCKETS=0, a
        model without ``dp_early_bucket``, a parameter order that does not match the flat layout)."""
        if (n <= 1 and not _force_exchange()) or os.environ.get("DREAM_DP_BUCKETS", "1") == "0":
            return None
        rep = self._replica(i)
        spec = rep.dp_early_bucket() if hasattr(rep, "dp_early_bucket") else None
        if spec is None:
            return None
        k, marker = spec
        if not (0 < k < len(offsets)) or any(offsets[j] < offsets[k] for j in range(k, len(offsets))) \
                or any(offsets[j] >= offsets[k] for j in range(k)):
            return None                                     # the bucket must be one contiguous tail of the flat buffer
        views = [gflat[o:o + m] for o, m in zip(offsets[k:], numels[k:])]
        return _EarlyBucket(k, marker, views, offsets[k])

    def _replica_backward(self, i, ctx, gos, gflat, offsets, numels, n=1):
        """One replica's backward; its parameter gradients land in ``gflat`` (the replica's flat gradient buffer).  -> the replica's
        early bucket when this step packed and marked one (its slice of ``gflat`` may be exchanged as soon as its event has passed)."""
        rep = self._replica(i)

        def run_and_pack(saved, grad_outs, early):
            grads = rep.dp_backward(saved, grad_outs, reducer=early) if early is not None else rep.dp_backward(saved, grad_outs)
            if early is not None and early.packed:          # the early bucket is in place already (and may be in flight): the rest only
                k = early.early_k
                self._pack_grads(grads[:k], gflat, offsets[:k], numels[:k])
            else:
                self._pack_grads(grads, gflat, offsets, numels)

        with torch.no_grad():
            if ctx[0] == "eager":
                from .models import _guarded_backward
                early = self._early_bucket(i, gflat, offsets, numels, n)
                if early is not None:
                    early.views = [v.view_as(p) for v, p in zip(early.views, rep.dp_parameters()[early.early_k:])]
                _guarded_backward(run_and_pack, ctx[1], gos, early)
                self.stats["eager"] += 1
                return early if early is not None and early.packed and early.marked else None
            entry = ctx[1]
            if entry["bwd"] is None:
                with DreamDataParallel._capture_lock:
                    entry["gos"] = [None if g is None else g.clone() for g in gos]
                    torch.cuda.current_stream().synchronize()
                    split = self.graph_split_leaves
                    if split is None:
                        split = 12
                    early = None
                    if split > 0:
                        devs = self.devices()
                        early = self._early_bucket(i, gflat, offsets, numels, n)      # (one backward graph: no early bucket -- an event
                        if early is not None:                                          #  recorded inside a capture cannot be waited for outside)
                            early.views = [v.view_as(p) for v, p in zip(early.views, rep.dp_parameters()[early.early_k:])]
                        graph = _SplitCapture(entry["pool"], split, gflat.device)
                        graph.capture(lambda: run_and_pack(entry["saved"], entry["gos"], early))
                    else:
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, pool=entry["pool"], capture_error_mode="thread_local"):
                            self._pack_grads(rep.dp_backward(entry["saved"], entry["gos"]), gflat, offsets, numels)
                    entry.update(bwd=graph, gflat_ptr=gflat.data_ptr(), early=early if early is not None and early.packed else None, n=n)
                    self.stats["captures"] += 1
            assert entry["gflat_ptr"] == gflat.data_ptr()
            for s, g in zip(entry["gos"], gos):
                if s is not None:
                    s.copy_(g, non_blocking=True)
            early = entry.get("early")
            if early is not None:
                early.marked = False
            entry["bwd"].replay()
            self.stats["replays"] += 1
            # (a graph captured for another number of devices keeps its plan: the bucket is then simply not exchanged early)
            return early if early is not None and early.marked and entry.get("n") == n else None

    @staticmethod
    def _pack_grads(grads, gflat, offsets, numels):
        views = [gflat[o:o + m].view_as(g) for o, m, g in zip(offsets, numels, grads)]
        torch._foreach_copy_(views, [g.contiguous() for g in grads])

    def _grad_buffer(self, i):
        """Replica i's persistent flat gradient buffer (laid out like its flat parameter buffer; the padding stays zero)."""
        rec = self._replica(i)._dream_flat
        if rec.get("grads") is None or rec["grads"].device != rec["params"].device:
            rec["grads"] = torch.zeros_like(rec["params"])
        return rec["grads"]

    # ---- the data-parallel step ---------------------------------------------------------------------------------------------
    def _forward_shards(self, x, save, post=None, post_key=None):
        """-> (outs[i] = list of output tensors of replica i (+ post(outs) appended when given), ctxs, chunk sizes)."""
        with self._lock:
            n = self.n_devices(x.shape[0])
            self._ensure_replicas(n)
            self._sync_replicas(n)
            xs = self._scatter(x, n)
            res = self._run([(lambda i=i: self._replica_forward(i, xs[i], save, post, post_key)) for i in range(len(xs))])
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
        splits = [g.split(sizes, dim=0) if g is not None else [None] * n for g in grad_outs]
        flats = [self._grad_buffer(i) for i in range(n)]
        self._release_grad_buffer(params, flats[0], offsets, numels)

        def job(i):
            def run():
                gos = [None if s[i] is None else (s[i].contiguous() if s[i].device == devs[i] else s[i].to(devs[i], non_blocking=True))
                       for s in splits]
                return self._replica_backward(i, ctxs[i], gos, flats[i], offsets, numels, n)
            return run
        with self._lock:
            _probe_sync("backward_begin")
            earlies = self._run([job(i) for i in range(n)])
            _probe_sync("backward_end")
            # The exchange: all-reduce(sum) over the replicas' flat gradient buffers, in place (RCCL over xGMI for distinct GPUs;
            # csrc/collective.hip).  Round 6: in TWO pieces where every replica marked its early bucket -- the tail gflat[lo:] on the
            # devices' exchange streams behind each replica's event (it overlaps what is left of the backward passes), then the head
            # behind the backward passes themselves; otherwise ONE call behind each replica's backward on that replica's stream.
            if n > 1 or _force_exchange():
                self._exchange(flats, earlies, devs[:n])
            object.__setattr__(self, "_reduced", n)
        total_flat = flats[0]
        # what step_replicas checks: any in-place edit of the gradients autograd is about to receive (clip_grad_norm_, manual
        # scaling: the views share the buffer's version counter) happens on the master only and must not be replayed blindly
        object.__setattr__(self, "_grad_version", total_flat._version)
        return [total_flat[o:o + m].view(prm.shape) for o, m, prm in zip(offsets, numels, params)]

    def _exchange(self, flats, earlies, devs):
        los = {e.lo for e in earlies if e is not None}
        if any(e is None for e in earlies) or len(los) != 1 or devs[0].type != "cuda":
            ops.allreduce_sum_(flats)
            self.stats["exchanges"] = self.stats.get("exchanges", 0) + 1
            return
        lo = los.pop()
        xs = []
        for d in devs:                                      # one exchange stream per device (replicas that share a device share it)
            if d.index not in self._xstreams:
                from . import _hip
                self._xstreams[d.index] = _hip.own_stream(d, "exchange")
            xs.append(self._xstreams[d.index])
        mains = [torch.cuda.current_stream(d) for d in devs]
        _probe_sync("exchange_begin")
        for x, e in zip(xs, earlies):
            x.wait_event(e.event)
        ops.allreduce_sum_([f[lo:] for f in flats], streams=xs)          # the early bucket: beside the rest of the backward passes
        _probe_sync("exchange_mid")
        for x, m in zip(xs, mains):
            x.wait_stream(m)
        if lo > 0:
            ops.allreduce_sum_([f[:lo] for f in flats], streams=xs)      # the late bucket: behind them
        for x, m in zip(xs, mains):
            m.wait_stream(x)
        _probe_sync("exchange_end")
        self.stats["exchanges"] = self.stats.get("exchanges", 0) + (2 if lo > 0 else 1)

    @staticmethod
    def _release_grad_buffer(params, flat, offsets, numels):
        """AccumulateGrad keeps the views this node returns as ``p.grad`` (it steals them when ``p.grad`` is None), so after
        ``zero_grad(set_to_none=False)`` or with gradient accumulation (two backwards, one step) the master's ``p.grad`` still
        aliases the persistent flat gradient buffer that the backward about to run overwrites in place -- ``p.grad += new`` would
        then add the buffer to itself.  Before anything is written the kept gradients move to a private flat copy with the same
        layout (one device copy, on this rare path only): autograd accumulates into the copy, the optimizer still sees ONE
        contiguous gradient buffer, and ``step_replicas`` notices that it is not the all-reduced buffer and lets the replicas
        be refreshed from the master instead."""
        lo = flat.data_ptr()
        hi = lo + 4 * flat.numel()
        held = [i for i, prm in enumerate(params) if prm.grad is not None and lo <= prm.grad.data_ptr() < hi]
        if not held:
            return
        with torch.no_grad():
            keep = flat.clone()
            for i in held:
                prm = params[i]
                off = (prm.grad.data_ptr() - lo) // 4
                prm.grad = keep[off:off + prm.grad.numel()].view(prm.grad.shape)

    # ---- optimizer hook: the identical update on every replica (dream_amd/optim.py) -------------------------------------------
    def step_replicas(self, kind, flat_params, flat_grads, hyper, master_state):
        """Called by HipAdam / HipSGD inside ``step()``, BEFORE the master's launch, with the slices of the master's flat
        parameter / gradient buffers they are about to update: applies the same update -- one launch per replica on the
        replica's device -- to the same slice of every replica's flat parameter buffer, with the replica's own copy of the
        all-reduced gradients and its own moment buffers (created from the master's current moments).  -> True when every replica was
        updated (the caller then confirms with ``mark_params_synced()`` after it bumped the master's versions)."""
        n = self._reduced
        object.__setattr__(self, "_reduced", 0)
        mrec = self.module._dream_flat
        if n <= 1 or n != len(self._replicas) + 1 or self._pstamp != _param_stamp(self.module) or mrec.get("grads") is None:
            return False
        if mrec["grads"]._version != self._grad_version:
            return False                                    # the master's gradients were edited after the all-reduce (clipping, scaling)
        lo, count = (flat_params.data_ptr() - mrec["params"].data_ptr()) // 4, int(flat_params.numel())
        if not (0 <= lo and lo + count <= mrec["params"].numel() and flat_grads.data_ptr() == mrec["grads"].data_ptr() + 4 * lo
                and int(flat_grads.numel()) == count):
            return False                                    # the optimizer is not stepping on this network's flat buffers
        devs = self.devices()
        with torch.no_grad():
            for i, rep in enumerate(self._replicas, 1):
                rec = rep._dream_flat
                p, g = rec["params"][lo:lo + count], rec["grads"][lo:lo + count]
                dev = devs[i]
                ctxm = torch.cuda.device(dev) if dev.type == "cuda" else _Null()
                with ctxm:
                    if kind == "adam":
                        st = self._opt_state.get(i)
                        if st is None or st[0].numel() != count:
                            st = self._opt_state[i] = tuple(m.to(dev, copy=True) for m in master_state)
                        ops.adam_step_(p, g, st[0], st[1], hyper["lr"], hyper["beta1"], hyper["beta2"], hyper["eps"], hyper["step"])
                    else:
                        ops.sgd_step_(p, g, hyper["lr"])
                for prm in rec["param_list"]:
                    ops.bump_version(prm)
        self.stats["replica_steps"] += len(self._replicas)
        return True

    def mark_params_synced(self):
        object.__setattr__(self, "_pstamp", _param_stamp(self.module))

    # ---- nn.Module interface ------------------------------------------------------------------------------------------------
    def forward(self, x, *args, **kwargs):
        if args or kwargs:
            return self.module(x, *args, **kwargs)
        one = self.n_devices(x.shape[0]) == 1
        if one and not (self.single_device_graphs and self.module.training and torch.is_grad_enabled()):
            return self.module(x)
        params = self.module.dp_parameters()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params) and self.module.dp_trainable():
            outs = list(_DataParallelFunction.apply(self, x, *params))
        elif one:
            return self.module(x)
        else:
            outs, _, _ = self._forward_shards(x, save=False)
            outs = self._gather(outs)
        return self.module.dp_finish(outs)

    def inference_shards(self, x, post, post_key=None):
        """No-grad forward of every chunk on its device followed by ``post(outputs)`` on the same device (the peak
        extraction of DreamNetwork.inference; ``post_key``: the settings ``post`` closes over, part of the hipGraph key).
        -> (outputs gathered on device_ids[0], [post result of each chunk])."""
        outs, _, _ = self._forward_shards(x, save=False, post=post, post_key=("post", post_key))
        return self._gather(outs, upto=len(outs[0]) - 1), [o[-1] for o in outs]


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _force_exchange():
    """DREAM_FORCE_RCCL=1: the exchange runs for a one-device list too (through RCCL itself: csrc/collective.hip) -- the one-GPU test box's
    way to execute the calls an 8-GPU node makes."""
    return os.environ.get("DREAM_FORCE_RCCL", "0") == "1"


def _distributed_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
