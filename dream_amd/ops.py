"""Thin functional layer over the C ABI: allocates outputs as torch tensors and forwards raw device
pointers + the current HIP stream to libdream_hip.so.  No arithmetic happens in Python/ATen here."""
import ctypes
import threading

import torch

from . import _hip
from ._hip import CONV_RELU, CONV_UPSAMPLE2X, CONV_OUT_NCHW, call, ptr, stream, stream_on

CONV_ZEROSTUFF2X = 8
CONV_POOL2 = 16
CONV_RELUMASK = 32
CONV_NO_KSPLIT = 128            # conv1x1: no K split over wavefronts (results independent of the row count in the last bit)
CONV_RES_AFTER_RELU = 64        # the residual is a skip connection: y = relu(conv + shift) + residual


def _f32(t):
    if t.dtype != torch.float32:
        raise RuntimeError("dream_amd: expected float32, got %s" % t.dtype)
    return t.contiguous()


def round_up(v, m):
    return (v + m - 1) // m * m


_pack_log = threading.local()
PACK_CONV1X1, PACK_WINOGRAD2, PACK_WINOGRAD4, PACK_CONVT_WINOGRAD2, PACK_CONVT_WINOGRAD4 = 0, 1, 2, 3, 4


class record_packs:
    """Context manager: collects (kind, weight, packed, cout, cin, mode) of the batchable weight packs made inside it (the first
    training step of a model; dream_pack_weights_batched then refreshes all of them with one launch per step)."""

    def __enter__(self):
        self.prev = getattr(_pack_log, "items", None)
        _pack_log.items = []
        return _pack_log.items

    def __exit__(self, *exc):
        _pack_log.items = self.prev


def _log_pack(kind, w, packed, cout, cin, mode):
    log = getattr(_pack_log, "items", None)
    if log is not None:
        log.append((kind, w, packed, cout, cin, mode))


def pack_job_table(descs, device):
    """Device-resident dream_pack_job table (include/dream_hip.h) for pack_weights_batched."""
    import numpy as np
    dt = np.dtype([("src", np.uint64), ("dst", np.uint64), ("kind", np.int32), ("cout", np.int32), ("cin", np.int32), ("mode", np.int32)])
    if int(_hip.lib().dream_pack_job_bytes()) != dt.itemsize:
        raise RuntimeError("dream_pack_job layout mismatch")
    arr = np.zeros(len(descs), dt)
    for i, (kind, w, packed, cout, cin, mode) in enumerate(descs):
        arr[i] = (ptr(w), ptr(packed), kind, cout, cin, mode)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


def pack_weights_batched(table, njobs, workgroups_per_job=16):
    call("dream_pack_weights_batched", ptr(table), int(njobs), int(workgroups_per_job), stream())


def pack_span_table(descs, device, floats_per_workgroup=1 << 15, max_parts=512):
    """Device-resident dream_pack_span table for pack_weights_spans: the workgroups of the one launch dealt out by the size of each job's
    packed copy (one per ``floats_per_workgroup`` floats, at least one) -> (table, number of workgroups)."""
    import numpy as np
    dt = np.dtype([("job", np.int32), ("part", np.int32), ("nparts", np.int32), ("reserved", np.int32)])
    if int(_hip.lib().dream_pack_span_bytes()) != dt.itemsize:
        raise RuntimeError("dream_pack_span layout mismatch")
    rows = []
    for job, d in enumerate(descs):
        nparts = max(1, min(int(max_parts), -(-int(d[2].numel()) // int(floats_per_workgroup))))
        rows.extend((job, part, nparts, 0) for part in range(nparts))
    arr = np.array(rows, dtype=dt)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device), len(rows)


def pack_weights_spans(table, spans, nspans):
    call("dream_pack_weights_spans", ptr(table), ptr(spans), int(nspans), stream())


class MultiCopyPlan:
    """Gathers a fixed list of fp32 tensors into fixed views of one flat buffer with ONE launch per call (csrc/elementwise.hip,
    dream_multi_copy_f32) -- what ``torch._foreach_copy_(views, tensors)`` does with one hipMemcpyAsync per tensor.  The
    destination table is built once (device-resident); a call ships the sources' pointers (one small pinned array)."""
    CHUNK = 1 << 16

    def __init__(self, views):
        import numpy as np
        dt = np.dtype([("dst", np.uint64), ("src_off", np.uint32), ("n", np.uint32), ("job", np.int32), ("reserved", np.int32)])
        if int(_hip.lib().dream_copy_chunk_bytes()) != dt.itemsize:
            raise RuntimeError("dream_copy_chunk layout mismatch")
        self.device = views[0].device
        self.numels = [int(v.numel()) for v in views]
        rows = []
        for job, v in enumerate(views):
            if v.dtype != torch.float32 or not v.is_contiguous() or v.device != self.device:
                raise RuntimeError("MultiCopyPlan: destinations must be contiguous fp32 views on one device")
            base, n = ptr(v), int(v.numel())
            for off in range(0, n, self.CHUNK):
                rows.append((base + 4 * off, off, min(self.CHUNK, n - off), job, 0))
        arr = np.array(rows, dtype=dt) if rows else np.zeros(0, dt)
        self.nchunks = len(rows)
        self.chunks = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)
        self.views = views                      # keeps the flat buffer alive
        # the pointer array is staged through a small ring of pinned host buffers: the host may run a step or more ahead of the
        # GPU, and a buffer must not be rewritten before its asynchronous copy has been consumed (an event per slot guards it)
        pin = self.device.type == "cuda"
        self._ring = [[torch.empty((len(views),), dtype=torch.int64, pin_memory=pin),
                       torch.empty((len(views),), dtype=torch.int64, device=self.device), None] for _ in range(4)]
        self._slot = 0

    def matches(self, tensors):
        return (len(tensors) == len(self.numels) and all(t is not None and t.dtype == torch.float32 and t.device == self.device
                and t.is_contiguous() and int(t.numel()) == n for t, n in zip(tensors, self.numels)))

    def run(self, tensors):
        slot = self._ring[self._slot]
        self._slot = (self._slot + 1) % len(self._ring)
        host, dev, done = slot
        if done is not None:
            done.synchronize()                      # (four launches ago: normally long complete)
        host.copy_(torch.tensor([ptr(t) for t in tensors], dtype=torch.int64))
        dev.copy_(host, non_blocking=True)
        call("dream_multi_copy_f32", ptr(dev), ptr(self.chunks), self.nchunks, stream())
        if self.device.type == "cuda":
            slot[2] = torch.cuda.Event()
            slot[2].record()


def pack_weight(w_oihw, mode=0):
    """OIHW [Cout,Cin,3,3] -> tap-major packed tensor (see dream_pack_conv3x3_weight).
    Returns (packed, rows, rows_pad, cols_pad)."""
    w = _f32(w_oihw)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    rows, cols = (cout, cin) if mode == 0 else (cin, cout)
    rows_pad, cols_pad = _hip.cout_pad(rows), round_up(cols, 16)
    packed = torch.empty((9, rows_pad, cols_pad), dtype=torch.float32, device=w.device)
    call("dream_pack_conv3x3_weight", ptr(w), ptr(packed), cout, cin, rows_pad, cols_pad, mode, stream())
    return packed, rows, rows_pad, cols_pad


def conv3x3(x_nhwc, packed, bias, cout, flags=0, relu_mask=None):
    """x: [B,Hs,Ws,Cin] NHWC (Cin == packed.shape[2]); returns [B,H,W,cout] (or NCHW with OUT_NCHW).
    relu_mask [B,H,W,cout]: y = relu_mask > 0 ? conv : 0 (data-gradient convs: the previous layer's ReLU gradient)."""
    if relu_mask is not None:
        return conv2d(x_nhwc, packed, cout, 3, 1, None, bias, relu_mask, flags | CONV_RELUMASK)
    x = _f32(x_nhwc)
    b, hs, ws, cin = (int(v) for v in x.shape)
    if cin != packed.shape[2]:
        raise RuntimeError("conv3x3: input has %d channels, packed weights expect %d" % (cin, packed.shape[2]))
    scale = 2 if flags & (CONV_UPSAMPLE2X | CONV_ZEROSTUFF2X) else 1
    h, w = hs * scale, ws * scale
    shape = (b, cout, h, w) if flags & CONV_OUT_NCHW else (b, h, w, cout)
    if flags & CONV_POOL2:
        shape = (b, h // 2, w // 2, cout)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    call("dream_conv3x3_nhwc_f32", ptr(x), ptr(packed), ptr(bias), ptr(y), b, h, w, cin, cout,
         int(packed.shape[1]), flags, stream())
    return y


def conv3x3_winograd4_pool_both(x_nhwc, packed_u, cout, shift=None, flags=0):
    """Training forward of a conv followed by MaxPool2d(2): -> (y_full [B,H,W,cout], y_pool [B,H/2,W/2,cout]) from one launch of the
    F(4x4,3x3) kernel (csrc/conv_wino4.hip MODE 4); flags: CONV_RELU.  Same contract on the input as conv3x3_winograd4."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    y = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    yp = torch.empty((b, h // 2, w // 2, cout), dtype=torch.float32, device=x.device)
    call("dream_conv3x3_winograd4_pool_both_nhwc_f32", ptr(x), ptr(packed_u), None, ptr(shift), ptr(y), ptr(yp), b, h, w, cin, cout,
         flags, stream())
    return y, yp


def pack_weight_winograd(w_oihw, mode=0):
    """OIHW [Cout,Cin,3,3] -> the transformed weights U = G g G^T of the Winograd F(2x2,3x3) kernel
    ([cols/16][16][rows_pad][16]).  mode 0: forward (rows = Cout); mode 1: data-gradient operator (rows = Cin).
    Returns (packed, rows)."""
    w = _f32(w_oihw)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    rows, cols = (cout, cin) if mode == 0 else (cin, cout)
    n = int(_hip.lib().dream_conv3x3_winograd_weight_floats(rows, cols))
    packed = torch.empty((n,), dtype=torch.float32, device=w.device)
    call("dream_pack_conv3x3_winograd_weight", ptr(w), ptr(packed), cout, cin, mode, stream())
    _log_pack(PACK_WINOGRAD2, w, packed, cout, cin, mode)
    return packed, rows


def conv3x3_winograd(x_nhwc, packed_u, cout, scale=None, shift=None, residual=None, flags=0):
    """3x3 stride-1 pad-1 conv by Winograd F(2x2,3x3): x [B,H,W,Cin] -> [B,H,W,cout] ([B,H/2,W/2,cout] with CONV_POOL2).
    flags: CONV_RELU | CONV_POOL2 | CONV_RELUMASK (residual = mask source)."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    shape = (b, h // 2, w // 2, cout) if flags & CONV_POOL2 else (b, h, w, cout)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    call("dream_conv3x3_winograd_nhwc_f32", ptr(x), ptr(packed_u), ptr(scale), ptr(shift), ptr(residual), ptr(y), b, h, w, cin,
         cout, flags, stream())
    return y


def winograd4_applies(cin, cout):
    """The F(4x4,3x3) kernel's two workgroup shapes: wide (more than 64 output channels: 128 per workgroup, 16-channel input
    chunks in pairs) and narrow (up to 64: 64 per workgroup, 8-channel chunks in pairs)."""
    if cout > 64:
        return cin % 32 == 0 and cin >= 64 and cout >= 128
    return cin % 16 == 0 and cin >= 32 and cout >= 48


def winograd_tile(h, w, cin, cout, batch=None):
    """Output tile of the Winograd kernel for a stride-1 3x3 conv on an h x w map: 4 = F(4x4,3x3) (csrc/conv_wino4.hip: 2.25
    multiplications per output, measured 1.3-1.4x F(2x2,3x3) on its layers, profiles/r03_microbench_wino4_b128.txt) where one of
    its workgroup shapes applies (winograd4_applies), rounding the map up to whole 4 x 4 tiles costs less than that gain, and the
    batch gives every CU at least two workgroups' worth of tiles; else 2 = F(2x2,3x3) (csrc/conv_wino.hip).
    DREAM_WINOGRAD_TILE=2 forces F(2x2)."""
    if _WINOGRAD_TILE_FORCED == 2 or not winograd4_applies(cin, cout):
        return 2
    if _WINOGRAD_TILE_FORCED == 4:
        return 4                       # tests: F(4x4) wherever the kernel applies, whatever the map and batch size
    t4 = ((h + 3) // 4) * ((w + 3) // 4)
    pad4 = t4 * 16
    pad2 = ((h + 1) // 2) * ((w + 1) // 2) * 4
    resident = 256 if cout > 64 else 512
    if batch is not None and ((batch * t4 + 15) // 16) * ((cout + 127) // 128) < 2 * resident:
        return 2                       # fewer than two 16-tile blocks per resident workgroup: the 32-tile F(2x2) blocks fill the chip better
    return 4 if pad4 < 1.25 * pad2 else 2


import os as _os
_WINOGRAD_TILE_FORCED = int(_os.environ.get("DREAM_WINOGRAD_TILE", "0"))


def set_winograd_tile(tile):
    """0: by layer (winograd_tile); 2: F(2x2,3x3) everywhere (A/B runs, bench.py --conv-algorithm winograd2); 4: F(4x4,3x3)
    wherever the kernel applies (parity tests on small batches)."""
    global _WINOGRAD_TILE_FORCED
    _WINOGRAD_TILE_FORCED = int(tile)


def pack_weight_winograd_tile(w_oihw, mode, tile):
    return pack_weight_winograd4(w_oihw, mode) if tile == 4 else pack_weight_winograd(w_oihw, mode)


def conv3x3_winograd_tile(tile, x_nhwc, packed_u, cout, scale=None, shift=None, residual=None, flags=0):
    fn = conv3x3_winograd4 if tile == 4 else conv3x3_winograd        # looked up at call time: bench.py wraps both
    return fn(x_nhwc, packed_u, cout, scale, shift, residual, flags)


def pack_weight_winograd4(w_oihw, mode=0):
    """OIHW [Cout,Cin,3,3] -> the transformed weights of the Winograd F(4x4,3x3) kernel ([cols/K][36][rows_pad][K], K = 16 or -- up to 64 rows -- 8); mode as
    pack_weight_winograd.  Returns (packed, rows)."""
    w = _f32(w_oihw)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    rows, cols = (cout, cin) if mode == 0 else (cin, cout)
    n = int(_hip.lib().dream_conv3x3_winograd4_weight_floats(rows, cols))
    packed = torch.empty((n,), dtype=torch.float32, device=w.device)
    call("dream_pack_conv3x3_winograd4_weight", ptr(w), ptr(packed), cout, cin, mode, stream())
    _log_pack(PACK_WINOGRAD4, w, packed, cout, cin, mode)
    return packed, rows


def conv3x3_winograd4(x_nhwc, packed_u, cout, scale=None, shift=None, residual=None, flags=0, out=None):
    """3x3 stride-1 pad-1 conv by Winograd F(4x4,3x3) (csrc/conv_wino4.hip): same contract as conv3x3_winograd; Cin % 32 == 0 (Cin % 16 == 0 for Cout <= 64).
    ``out``: a contiguous tensor of the result's shape to write into (a batch slice of a larger tensor)."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    shape = (b, h // 2, w // 2, cout) if flags & CONV_POOL2 else (b, h, w, cout)
    if out is not None:
        if tuple(out.shape) != shape or not out.is_contiguous() or out.dtype != torch.float32 or out.device != x.device:
            raise RuntimeError("conv3x3_winograd4: out must be a contiguous float32 tensor of shape %s on %s" % (shape, x.device))
        y = out
    else:
        y = torch.empty(shape, dtype=torch.float32, device=x.device)
    call("dream_conv3x3_winograd4_nhwc_f32", ptr(x), ptr(packed_u), ptr(scale), ptr(shift), ptr(residual), ptr(y), b, h, w, cin,
         cout, flags, stream())
    return y


def pack_convT4x4_winograd_weight(wT, mode=0):
    """[Cin,Cout,4,4] ConvTranspose2d(k4,s2,p1) weight -> the four phases' Winograd-transformed 3x3 kernels of the forward
    operator (mode 0; returns (u4, cout)) or of the data-gradient operator (mode 1; returns (u4, cin))."""
    w = _f32(wT)
    cin, cout = int(w.shape[0]), int(w.shape[1])
    rows, cols = (cout, cin) if mode == 0 else (cin, cout)
    n = int(_hip.lib().dream_convT4x4_winograd_weight_floats(rows, cols))
    u4 = torch.empty(n, dtype=torch.float32, device=w.device)
    scratch = torch.empty(4 * cout * cin * 9, dtype=torch.float32, device=w.device)
    call("dream_pack_convT4x4_winograd_weight", ptr(w), ptr(u4), ptr(scratch), cin, cout, mode, stream())
    for ph in range(4):                                  # what a batched re-pack needs: one job per phase, straight from wT
        _log_pack(PACK_CONVT_WINOGRAD2, w, u4[ph * (n // 4):(ph + 1) * (n // 4)], rows, cols, ph | (mode << 2))
    return u4, rows


def conv4x4s2_winograd(dy_nhwc, u4_mode1, cin):
    """Data gradient of ConvTranspose2d(k4,s2,p1): dy [B,2H,2W,Cout] -> dx [B,H,W,cin] (the four phase convs, summed)."""
    dy = _f32(dy_nhwc)
    b, h2, w2, cout = (int(v) for v in dy.shape)
    dx = torch.empty((b, h2 // 2, w2 // 2, cin), dtype=torch.float32, device=dy.device)
    call("dream_conv4x4s2_winograd_nhwc_f32", ptr(dy), ptr(u4_mode1), ptr(dx), b, h2 // 2, w2 // 2, cout, cin, stream())
    return dx


def convT4x4_winograd_applies(x_nhwc, cout):
    cin = int(x_nhwc.shape[3])
    return cin % 16 == 0 and cin >= 32 and cout > 64


def conv_transpose4x4s2_winograd(x_nhwc, u4, cout, scale=None, shift=None, flags=0, direct_taps=16):
    """ConvTranspose2d(k4,s2,p1) * scale + shift (ReLU) by minimal filtering (9/16 of the direct multiplications), NHWC."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    y = torch.empty((b, 2 * h, 2 * w, cout), dtype=torch.float32, device=x.device)
    call("dream_conv_transpose4x4s2_winograd_nhwc_f32", ptr(x), ptr(u4), ptr(scale), ptr(shift), ptr(y), b, h, w, cin, cout, flags, stream())
    return y


def convT4x4_winograd_tile(x_nhwc, cout):
    """Output tile of the Winograd kernel for ConvTranspose2d(k4,s2,p1) on this input: 4 = the F(4x4,3x3) kernel with the 25-position
    phase patterns (wide workgroup shape only), 2 = the F(2x2,3x3) kernel with its 9-position patterns."""
    b, h, w, cin = (int(v) for v in x_nhwc.shape)
    return _convT4x4_tile(b, h, w, cin, cout)


def _convT4x4_tile(b, h, w, cin, cout):
    if cout < (128 if _WINOGRAD_TILE_FORCED == 4 else 256) or winograd_tile(h, w, cin, cout, b) != 4:
        return 2                       # (128 output channels = one channel block per tile block: measured slower than F(2x2) in vgg_q / vgg_f)
    if _WINOGRAD_TILE_FORCED == 4:
        return 4
    # measured (profiles/r03_microbench_convT4.txt, 256 -> 256): 1.21x the F(2x2) form on maps of 100 x 100 and more, 1.05-1.13x on
    # 50 x 50, slower on 13 x 13 (the 25-position kernel runs at 0.62 of the MFMA peak, the 9-position one at 0.73)
    return 4 if h * w >= 48 * 48 else 2


def pack_convT4x4_winograd_weight_tile(wT, tile, mode=0):
    return pack_convT4x4_winograd4_weight(wT, mode) if tile == 4 else pack_convT4x4_winograd_weight(wT, mode)


def conv4x4s2_winograd_tile(tile, dy_nhwc, u4_mode1, cin):
    fn = conv4x4s2_winograd4 if tile == 4 else conv4x4s2_winograd            # looked up at call time: bench.py wraps both
    return fn(dy_nhwc, u4_mode1, cin)


def conv4x4s2_winograd_tile_of(dy_nhwc, cin):
    """Tile of the data gradient of ConvTranspose2d(k4,s2,p1) (dy [B,2H,2W,Cout] -> dx [B,H,W,cin]): the forward rule on the
    gradient's own conv (map H x W, cin output channels)."""
    b, h2, w2, cout = (int(v) for v in dy_nhwc.shape)
    # measured (profiles/r03_microbench_convT4.txt): no gain over the 9-position F(2x2) form on any map (1.00x at 200x200, 0.88-0.96x
    # below): the gradient reads stride-2 phase views of dy -- a 6x6 patch spans 12x12 stored pixels -- and the 1.44x fewer
    # multiplications do not pay for that.  The F(4x4) form stays reachable for the tests (set_winograd_tile(4)).
    if _WINOGRAD_TILE_FORCED != 4:
        return 2
    return _convT4x4_tile(b, h2 // 2, w2 // 2, cout, cin)        # the gradient's conv: cout input channels, cin output channels


def conv_transpose4x4s2_winograd_tile(tile, x_nhwc, u4, cout, scale=None, shift=None, flags=0, direct_taps=16):
    fn = conv_transpose4x4s2_winograd4 if tile == 4 else conv_transpose4x4s2_winograd      # looked up at call time: bench.py wraps both
    return fn(x_nhwc, u4, cout, scale, shift, flags, direct_taps=direct_taps)


def pack_convT4x4_winograd4_weight(wT, mode=0):
    """[Cin,Cout,4,4] ConvTranspose2d(k4,s2,p1) weight -> the four phases' F(4x4,3x3)-transformed zero-padded 3x3 kernels of the
    forward operator (mode 0; returns (u4, cout)) or of the data-gradient operator (mode 1; returns (u4, cin))."""
    w = _f32(wT)
    cin, cout = int(w.shape[0]), int(w.shape[1])
    rows, cols = (cout, cin) if mode == 0 else (cin, cout)
    n = int(_hip.lib().dream_convT4x4_winograd4_weight_floats(rows, cols))
    u4 = torch.empty(n, dtype=torch.float32, device=w.device)
    scratch = torch.empty(4 * cout * cin * 9, dtype=torch.float32, device=w.device)
    call("dream_pack_convT4x4_winograd4_weight", ptr(w), ptr(u4), ptr(scratch), cin, cout, mode, stream())
    for ph in range(4):
        _log_pack(PACK_CONVT_WINOGRAD4, w, u4[ph * (n // 4):(ph + 1) * (n // 4)], rows, cols, ph | (mode << 2))
    return u4, rows


def conv4x4s2_winograd4(dy_nhwc, u4_mode1, cin):
    """Data gradient of ConvTranspose2d(k4,s2,p1) on the F(4x4) kernel's 25-position patterns: dy [B,2H,2W,Cout] -> dx [B,H,W,cin]."""
    dy = _f32(dy_nhwc)
    b, h2, w2, cout = (int(v) for v in dy.shape)
    dx = torch.empty((b, h2 // 2, w2 // 2, cin), dtype=torch.float32, device=dy.device)
    call("dream_conv4x4s2_winograd4_nhwc_f32", ptr(dy), ptr(u4_mode1), ptr(dx), b, h2 // 2, w2 // 2, cout, cin, stream())
    return dx


def conv_transpose4x4s2_winograd4(x_nhwc, u4, cout, scale=None, shift=None, flags=0, direct_taps=16):
    """ConvTranspose2d(k4,s2,p1) * scale + shift (ReLU) by minimal filtering on the F(4x4) kernel (25/64 of the direct
    multiplications), NHWC; cout > 64, Cin a multiple of 32."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    y = torch.empty((b, 2 * h, 2 * w, cout), dtype=torch.float32, device=x.device)
    call("dream_conv_transpose4x4s2_winograd4_nhwc_f32", ptr(x), ptr(u4), ptr(scale), ptr(shift), ptr(y), b, h, w, cin, cout, flags, stream())
    return y


def conv3x3_first(x_nchw, w_oihw, bias, relu=True):
    x, w = _f32(x_nchw), _f32(w_oihw)
    b, cin, h, wd = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    y = torch.empty((b, h, wd, cout), dtype=torch.float32, device=x.device)
    call("dream_conv3x3_first_nchw_f32", ptr(x), ptr(w), ptr(bias), ptr(y), b, h, wd, cin, cout,
         1 if relu else 0, stream())
    return y


def conv_transpose3x3s2(x_nhwc, packed_mode1, bias, cout, relu=True, skip=None):
    """ConvTranspose2d(k3,s2,p1,output_padding 1) (+ReLU) by sub-pixel phases: [B,H,W,Cin] -> [B,2H,2W,cout].
    packed_mode1 = pack_weight(convT.weight, 1)[0] (the packing the zero-stuffed form uses).  ``skip`` [B,2H,2W,cout]: a skip
    connection added after the ReLU in the same epilogue (dream/models.py:796-799)."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    if cin != packed_mode1.shape[2]:
        raise RuntimeError("conv_transpose3x3s2: input has %d channels, packed weights expect %d" % (cin, packed_mode1.shape[2]))
    y = torch.empty((b, 2 * h, 2 * w, cout), dtype=torch.float32, device=x.device)
    if skip is not None:
        if tuple(skip.shape) != tuple(y.shape):
            raise RuntimeError("conv_transpose3x3s2: skip shape %s != output shape %s" % (tuple(skip.shape), tuple(y.shape)))
        call("dream_conv_transpose3x3s2_res_nhwc_f32", ptr(x), ptr(packed_mode1), ptr(bias), ptr(_f32(skip)), ptr(y), b, h, w, cin, cout,
             int(packed_mode1.shape[1]), (CONV_RELU if relu else 0) | CONV_RES_AFTER_RELU, stream())
        return y
    call("dream_conv_transpose3x3s2_nhwc_f32", ptr(x), ptr(packed_mode1), ptr(bias), ptr(y), b, h, w, cin, cout,
         int(packed_mode1.shape[1]), CONV_RELU if relu else 0, stream())
    return y


def maxpool2(x_nhwc):
    x = _f32(x_nhwc)
    b, h, w, c = (int(v) for v in x.shape)
    y = torch.empty((b, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    call("dream_maxpool2_nhwc_f32", ptr(x), ptr(y), b, h, w, c, stream())
    return y


def nchw_to_nhwc(x, cpad=None):
    x = _f32(x)
    b, c, h, w = (int(v) for v in x.shape)
    if cpad is None or cpad == c:
        y = torch.empty((b, h, w, c), dtype=torch.float32, device=x.device)
        call("dream_nchw_to_nhwc_f32", ptr(x), ptr(y), b, c, h, w, stream())
    else:
        y = torch.empty((b, h, w, cpad), dtype=torch.float32, device=x.device)
        call("dream_nchw_to_nhwc_pad_f32", ptr(x), ptr(y), b, c, h, w, cpad, stream())
    return y


def nhwc_to_nchw(x):
    x = _f32(x)
    b, h, w, c = (int(v) for v in x.shape)
    y = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
    call("dream_nhwc_to_nchw_f32", ptr(x), ptr(y), b, c, h, w, stream())
    return y


# ---- peak extraction --------------------------------------------------------------------------------
def keypoints_from_belief_maps(maps_bkhw, offset, use_peak_scores=True, next_best_score=0.25):
    """[B,K,H,W] device fp32 -> ([B,K,2] fp32 device tensor, [B,K] int32 peak counts).  use_peak_scores / next_best_score:
    DreamNetwork.use_belief_peak_scores / .belief_peak_next_best_score (dream/network.py:189-191,553-560)."""
    m = _f32(maps_bkhw)
    b, k, h, w = (int(v) for v in m.shape)
    scratch = torch.empty((2, b * k, h, w), dtype=torch.float32, device=m.device)
    kps = torch.empty((b, k, 2), dtype=torch.float32, device=m.device)
    counts = torch.empty((b, k), dtype=torch.int32, device=m.device)
    call("dream_keypoints_from_belief_maps_rule_f32", ptr(m), ptr(scratch), ptr(kps), ptr(counts), b * k, h, w,
         float(offset), 1 if use_peak_scores else 0, float(next_best_score), stream())
    return kps, counts


def peaks_list(maps_khw, offset, cap=256):
    """[K,H,W] -> (xy [K,cap,2] fp64, score [K,cap] fp32, counts [K] int32), all on the device."""
    m = _f32(maps_khw)
    k, h, w = (int(v) for v in m.shape)
    while True:
        scratch = torch.empty((2, k, h, w), dtype=torch.float32, device=m.device)
        xy = torch.empty((k, cap, 2), dtype=torch.float64, device=m.device)
        score = torch.empty((k, cap), dtype=torch.float32, device=m.device)
        counts = torch.empty((k,), dtype=torch.int32, device=m.device)
        call("dream_peaks_from_belief_maps_f32", ptr(m), ptr(scratch), ptr(xy), ptr(score), ptr(counts), k, h, w,
             cap, float(offset), stream())
        most = int(counts.max().item()) if k else 0
        if most <= cap:
            return xy, score, counts
        cap = round_up(most, 256)


def gaussian_sigma3(maps_nhw):
    m = _f32(maps_nhw)
    n, h, w = (int(v) for v in m.shape)
    tmp, out = torch.empty_like(m), torch.empty_like(m)
    call("dream_gaussian_sigma3_f32", ptr(m), ptr(tmp), ptr(out), n, h, w, stream())
    return out


def softargmax(maps_bkhw, beta, size_mult=1.0):
    m = _f32(maps_bkhw)
    b, k, h, w = (int(v) for v in m.shape)
    scratch = torch.empty_like(m)
    out = torch.empty((b, k, 2), dtype=torch.float32, device=m.device)
    call("dream_softargmax_f32", ptr(m), ptr(_f32(beta)), ptr(scratch), ptr(out), b * k, k, h, w, float(size_mult),
         stream())
    return out


# ---- training operators -------------------------------------------------------------------------------
def mse_fwd_bwd(out, target, want_grad=True, kind="mse"):
    """Returns (loss 0-dim tensor, grad or None): mean((o-t)^2) and 2(o-t)/N  (kind "mse"), or the SmoothL1 (beta 1)
    mean loss and its gradient (kind "huber")."""
    o, t = _f32(out), _f32(target)
    if o.shape != t.shape:
        raise RuntimeError("loss: shape mismatch %s vs %s" % (tuple(o.shape), tuple(t.shape)))
    n = o.numel()
    loss_sum = torch.empty((1,), dtype=torch.float32, device=o.device)
    ws = _workspace(int(_hip.lib().dream_loss_workspace(n)), o.device)
    grad = torch.empty_like(o) if want_grad else None
    fn = "dream_mse_fwd_bwd_f32" if kind == "mse" else "dream_smoothl1_fwd_bwd_f32"
    call(fn, ptr(o), ptr(t), ptr(grad), ptr(loss_sum), ptr(ws), n, float(n), stream())
    return loss_sum[0] / n, grad


def relu_bwd_(dy, y):
    """in place: dy *= (y > 0)"""
    call("dream_relu_bwd_f32", ptr(dy), ptr(y), ptr(dy), dy.numel(), stream())
    return dy


def maxpool2_bwd(dy, x, relu=False):
    """relu: x is a ReLU output and the gradient continues through that ReLU (dx *= x > 0) in the same pass."""
    b, h, w, c = (int(v) for v in x.shape)
    dx = torch.empty_like(x)
    call("dream_maxpool2_relu_bwd_nhwc_f32" if relu else "dream_maxpool2_bwd_nhwc_f32", ptr(_f32(dy)), ptr(x), ptr(dx),
         b, h, w, c, stream())
    return dx


def upsample2_bwd(dy):
    b, h, w, c = (int(v) for v in dy.shape)
    dx = torch.empty((b, h // 2, w // 2, c), dtype=torch.float32, device=dy.device)
    call("dream_upsample2_bwd_nhwc_f32", ptr(_f32(dy)), ptr(dx), b, h, w, c, stream())
    return dx


def subsample2(x):
    """x[:, ::2, ::2, :] of an NHWC tensor as a contiguous tensor: the pixels a stride-2 1x1 conv reads."""
    x = _f32(x)
    b, h, w, c = (int(v) for v in x.shape)
    y = torch.empty((b, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.float32, device=x.device)
    call("dream_subsample2_nhwc_f32", ptr(x), ptr(y), b, h, w, c, stream())
    return y


def scatter2(ys, h, w):
    """The transpose of subsample2: an NHWC tensor of height h and width w with ys at its even pixels and zeros elsewhere."""
    ys = _f32(ys)
    b, hs, ws, c = (int(v) for v in ys.shape)
    if (hs, ws) != ((h + 1) // 2, (w + 1) // 2):
        raise RuntimeError("scatter2: %dx%d is not the subsampled grid of %dx%d" % (hs, ws, h, w))
    x = torch.empty((b, h, w, c), dtype=torch.float32, device=ys.device)
    call("dream_scatter2_nhwc_f32", ptr(ys), ptr(x), b, h, w, c, stream())
    return x


def im2col3s2(x):
    """The patches of a 3x3 stride-2 pad-1 conv as rows: [B, Ho, Wo, 9 C], column t C + c = tap t = 3 ky + kx of channel c."""
    x = _f32(x)
    b, h, w, c = (int(v) for v in x.shape)
    col = torch.empty((b, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 9 * c), dtype=torch.float32, device=x.device)
    call("dream_im2col3s2_nhwc_f32", ptr(x), ptr(col), b, h, w, c, stream())
    return col


def col2im3s2(col, h, w):
    """The transpose of im2col3s2: [B, h, w, C] with every pixel the sum of the patch entries that read it."""
    col = _f32(col)
    b, ho, wo, c9 = (int(v) for v in col.shape)
    if (ho, wo) != ((h - 1) // 2 + 1, (w - 1) // 2 + 1) or c9 % 9:
        raise RuntimeError("col2im3s2: %dx%dx%d are not the patch rows of a %dx%d map" % (ho, wo, c9, h, w))
    dx = torch.empty((b, h, w, c9 // 9), dtype=torch.float32, device=col.device)
    call("dream_col2im3s2_nhwc_f32", ptr(col), ptr(dx), b, h, w, c9 // 9, stream())
    return dx


def col2im4s2(g, cout, bias=None, scale=None, flags=0):
    """[B,H,W,16 cout] tap contributions of a ConvTranspose2d(k4, s2, p1) (column (4 ky + kx) cout + c) -> its output [B,2H,2W,cout]:
    bias + sum (scale None), or sum * scale + bias (the folded BatchNorm of the evaluation path); flags: CONV_RELU."""
    g = _f32(g)
    b, h, w, n = (int(v) for v in g.shape)
    if n != 16 * cout:
        raise RuntimeError("col2im4s2: %d columns are not 16 x %d" % (n, cout))
    z = torch.empty((b, 2 * h, 2 * w, cout), dtype=torch.float32, device=g.device)
    call("dream_col2im4s2_nhwc_f32", ptr(g), ptr(None if scale is None else _f32(scale)), ptr(None if bias is None else _f32(bias)), ptr(z),
         b, h, w, cout, flags, stream())
    return z


class wgrad_width:
    """``with ops.wgrad_width(p):`` the weight-gradient launches planned by this thread inside the block split their contraction only
    until ``p`` per cent of the full-width workgroup count exist (csrc/api.hip dream_wgrad_set_width: thread-local) -- for leaves that
    run on a second stream beside the data-gradient chain.  100 restores the full width."""

    def __init__(self, percent):
        self.percent = int(percent)

    def __enter__(self):
        if self.percent != 100:
            call("dream_wgrad_set_width", self.percent)
        return self

    def __exit__(self, *exc):
        if self.percent != 100:
            call("dream_wgrad_set_width", 100)
        return False


# width of the weight-gradient launches that run on the second stream (models._SideStream), per cent of the chip
SIDE_WGRAD_WIDTH = int(_os.environ.get("DREAM_SIDE_WGRAD_WIDTH", "100"))


def conv3x3_wgrad(x_nhwc, dy_nhwc, cout, cin, flags=0):
    """-> (dW OIHW [cout,cin,3,3], dbias [cout]).  dy may carry padded channels (>= cout)."""
    x, dy = _f32(x_nhwc), _f32(dy_nhwc)
    b, h, w, cdy = (int(v) for v in dy.shape)
    if int(x.shape[3]) != cin:
        raise RuntimeError("wgrad: x has %d channels, expected %d" % (x.shape[3], cin))
    rows_pad = round_up(cdy, 64)
    nbytes = int(_hip.lib().dream_conv3x3_wgrad_workspace(b, h, w, cin, rows_pad))
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    dwp = torch.empty((9, rows_pad, cin), dtype=torch.float32, device=x.device)
    dbias = torch.empty((cdy,), dtype=torch.float32, device=x.device)
    call("dream_conv3x3_wgrad_nhwc_f32", ptr(x), ptr(dy), ptr(dwp), ptr(dbias), ptr(ws), b, h, w, cin, cdy, rows_pad,
         flags, stream())
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    call("dream_unpack_conv3x3_weight", ptr(dwp), ptr(dw), cout, cin, rows_pad, cin, stream())
    return dw, dbias[:cout].contiguous()


# A/B switch: DREAM_WGRAD_BIAS_FUSION=0 restores the stand-alone channel_sum pass for the bias gradients
WGRAD_BIAS_FUSION = _os.environ.get("DREAM_WGRAD_BIAS_FUSION", "1") != "0"


# DREAM_UPS_WGRAD=convT9: the weight gradient of an upsample + conv3x3 on the nine-position transposed-conv form instead of the
# sixteen-position kernel with the fused upsample.  Measured round 6 (profiles/r06_ab_ups_wgrad_convT9.txt, vgg_q training b=128,
# alternating): 701.1 vs 700.9 frames/s -- 1.64x fewer MFMAs at 0.54 instead of 0.70 of the peak: no gain, so the default stays.
UPS_WGRAD_AS_CONVT = _os.environ.get("DREAM_UPS_WGRAD", "winograd16") == "convT9"


WGRAD_WINOGRAD_MIN_PIXELS_64 = int(__import__("os").environ.get("DREAM_WGRAD_WINOGRAD_MIN_PIXELS", "150000"))      # A/B: 4000000 = the rule of rounds 2-5


def wgrad_winograd_pays(pixels, cin, cout):
    """Where the Winograd-domain weight gradient beats the direct kernel (profiles/r02_microbench_wgrad_wino_b128.txt:
    1.2-2.1x on the layers with >= 128 x 64 channel pairs; on 64 x 64-channel layers, whose split-K partials and short loops weigh
    more, from ~150 k pixels on since the round-6 kernel: 16 x 100 x 100 pixels 0.178 -> 0.103 ms, 128 x 100 x 100 1.4 -> 0.43 ms,
    gpurun_out r06_wgw_b16 / profiles/r06_ab_convT_wgrad_five_loads.txt; rounds 2-5: from 4 M pixels on)."""
    return cin % 64 == 0 and cout % 16 == 0 and cout >= 64 and (cin * cout >= 8192 or pixels >= WGRAD_WINOGRAD_MIN_PIXELS_64)


def conv3x3_wgrad_winograd(x_nhwc, dy_nhwc, cout, cin, want_bias=True, flags=0):
    """Weight (+ bias) gradient of a 3x3 stride-1 conv in the Winograd F(2x2,3x3) domain -> (dW OIHW [cout,cin,3,3], dbias
    [cout] or None).  cin % 64 == 0, cout % 16 == 0; dy may carry padded channels (>= cout).  flags: CONV_UPSAMPLE2X when x is
    the half-resolution input of a conv that followed a nearest x2 upsample (channels a multiple of 64).  The bias gradient
    is summed in the kernel's dy loader where the LDS kernel runs (channels multiples of 64; WGRAD_BIAS_FUSION), by a
    channel_sum pass elsewhere."""
    x, dy = _f32(x_nhwc), _f32(dy_nhwc)
    b, h, w, cdy = (int(v) for v in dy.shape)
    if int(x.shape[3]) != cin:
        raise RuntimeError("wgrad: x has %d channels, expected %d" % (x.shape[3], cin))
    lib = _hip.lib()
    nbytes = int(lib.dream_conv3x3_wgrad_winograd_workspace(b, h, w, cin, cout))
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    db = None
    if want_bias and WGRAD_BIAS_FUSION and lib.dream_conv3x3_wgrad_winograd_fuses_bias(cin, cout, cdy):
        db = torch.empty((cout,), dtype=torch.float32, device=x.device)
    call("dream_conv3x3_wgrad_winograd_bias_nhwc_f32", ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(ws), b, h, w, cin, cout, cdy, flags,
         stream())
    if want_bias and db is None:
        db = channel_sum(dy)[:cout].contiguous()
    return dw, db


def conv3x3_first_wgrad(x_nchw, dy_nhwc):
    x, dy = _f32(x_nchw), _f32(dy_nhwc)
    b, cin, h, w = (int(v) for v in x.shape)
    cout = int(dy.shape[3])
    nbytes = int(_hip.lib().dream_conv3x3_first_wgrad_workspace(b, h, w, cin, cout))
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device)
    call("dream_conv3x3_first_wgrad_f32", ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(ws), nbytes, b, h, w, cin, cout,
         stream())
    return dw, db


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, step):
    call("dream_adam_step_f32", ptr(p), ptr(_f32(g)), ptr(m), ptr(v), p.numel(), float(lr), float(beta1), float(beta2),
         float(eps), int(step), stream())


def sgd_step_(p, g, lr):
    call("dream_sgd_step_f32", ptr(p), ptr(_f32(g)), p.numel(), float(lr), stream())


# ---- generic conv path (ResNet) ---------------------------------------------------------------------
def pack_conv_weight(w_oihw, mode=0):
    """OIHW [Cout,Cin,kh,kw] (kh*kw taps) -> [ntaps][rows_pad][cols_pad]; returns (packed, rows, ntaps)."""
    w = _f32(w_oihw)
    cout, cin, kh, kw = (int(v) for v in w.shape)
    rows, cols = (cout, cin) if mode == 0 else (cin, cout)
    rows_pad, cols_pad = _hip.cout_pad(rows), round_up(cols, 16)
    packed = torch.empty((kh * kw, rows_pad, cols_pad), dtype=torch.float32, device=w.device)
    call("dream_pack_conv_weight", ptr(w), ptr(packed), cout, cin, kh * kw, rows_pad, cols_pad, mode, stream())
    return packed, rows, kh * kw


def pack_matrix_weight(w2d, cols_pad):
    """[Cout, K] matrix (e.g. the 7x7 stem weight flattened to K = 147) -> 1-tap packed [1][rows_pad][cols_pad]."""
    w = _f32(w2d)
    cout, k = (int(v) for v in w.shape)
    rows_pad = _hip.cout_pad(cout)
    packed = torch.empty((1, rows_pad, cols_pad), dtype=torch.float32, device=w.device)
    call("dream_pack_conv_weight", ptr(w), ptr(packed), cout, k, 1, rows_pad, cols_pad, 0, stream())
    return packed, cout, 1


def pack_convT4x4_weight(wT):
    """ConvTranspose2d weight [Cin,Cout,4,4] -> [4][4][rows_pad][cols_pad]; returns (packed, cout)."""
    w = _f32(wT)
    cin, cout = int(w.shape[0]), int(w.shape[1])
    rows_pad, cols_pad = _hip.cout_pad(cout), round_up(cin, 16)
    packed = torch.empty((4, 4, rows_pad, cols_pad), dtype=torch.float32, device=w.device)
    call("dream_pack_convT4x4_weight", ptr(w), ptr(packed), cin, cout, rows_pad, cols_pad, stream())
    return packed, cout


def upsample_conv_weight(w_oihw):
    """Conv2d(k3,p1) weight [Cout,Cin,3,3] applied after a nearest x2 upsample -> the equivalent ConvTranspose2d(k4,s2,p1)
    weight [Cin,Cout,4,4] (see dream_upsample_conv3x3_weight_as_convT4x4)."""
    w = _f32(w_oihw)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    wt4 = torch.empty((cin, cout, 4, 4), dtype=torch.float32, device=w.device)
    call("dream_upsample_conv3x3_weight_as_convT4x4", ptr(w), ptr(wt4), cout, cin, stream())
    return wt4


def conv2d(x_nhwc, packed, cout, ksize, stride=1, scale=None, shift=None, residual=None, flags=0):
    """k x k (1 or 3) conv, stride 1/2, pad k/2, fused y = conv*scale + shift (+residual) (ReLU)."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    if cin != packed.shape[-1]:
        raise RuntimeError("conv2d: input has %d channels, packed weights expect %d" % (cin, packed.shape[-1]))
    if flags & (CONV_UPSAMPLE2X | CONV_ZEROSTUFF2X):
        h, w = 2 * h, 2 * w
    pad = ksize // 2
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    shape = (b, cout, ho, wo) if flags & CONV_OUT_NCHW else (b, ho, wo, cout)
    if flags & CONV_POOL2:
        shape = (b, ho // 2, wo // 2, cout)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    if residual is not None and tuple(residual.shape) != shape:
        raise RuntimeError("conv2d: residual shape %s != output shape %s" % (tuple(residual.shape), shape))
    call("dream_conv2d_nhwc_f32", ptr(x), ptr(packed), ptr(scale), ptr(shift), ptr(residual), ptr(y), b, h, w, cin,
         cout, int(packed.shape[-2]), ksize, stride, flags, stream())
    return y


def pack_conv1x1_weight(w_oihw, mode=0):
    """[Cout,Cin,1,1] -> the LDS-free GEMM kernel's operand layout; mode 1: the data-gradient operator.  Returns (packed, rows)."""
    w = _f32(w_oihw)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    rows, k = (cout, cin) if mode == 0 else (cin, cout)
    n = int(_hip.lib().dream_conv1x1_weight_floats(rows, k))
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    call("dream_pack_conv1x1_weight", ptr(w), ptr(packed), cout, cin, mode, stream())
    _log_pack(PACK_CONV1X1, w, packed, cout, cin, mode)
    return packed, rows


def conv1x1_applies(x_nhwc, cout):
    """The LDS-free GEMM kernel takes stride-1 1x1 convs with K % 32 == 0, N % 4 == 0 and tensors below 2 GB."""
    k = int(x_nhwc.shape[3])
    return k % 32 == 0 and cout % 4 == 0 and x_nhwc.numel() * 4 < (1 << 31)


def conv1x1(x_nhwc, packed, cout, scale=None, shift=None, residual=None, flags=0):
    """1x1 stride-1 conv, y = conv*scale + shift (+residual) (ReLU), NHWC."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    y = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise RuntimeError("conv1x1: residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    call("dream_conv1x1_nhwc_f32", ptr(x), ptr(packed), ptr(scale), ptr(shift), ptr(residual), ptr(y), b * h * w, cin, cout, cin,
         flags, stream())
    return y


def conv1x1_pre(x_nhwc, packed, cout, pre_ab, shift=None):
    """1x1 conv of relu(pre_ab[0] x + pre_ab[1]) (a train-mode BatchNorm + ReLU applied in the loader: its output is never stored), + shift."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    y = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    call("dream_conv1x1_pre_nhwc_f32", ptr(x), ptr(packed), ptr(pre_ab), ptr(shift), ptr(y), b * h * w, cin, cout, cin, stream())
    return y


def conv1x1_wgrad_applies(x_nhwc, dy_nhwc, cout):
    cin, cdy = int(x_nhwc.shape[3]), int(dy_nhwc.shape[3])
    # (measured: 1.2-1.8x the direct weight-gradient kernel from 64 x 256 channel pairs on, 0.74x at 64 x 64)
    return (cin % 64 == 0 and cout % 4 == 0 and cdy % 4 == 0 and cin * cout >= 8192 and tuple(x_nhwc.shape[:3]) == tuple(dy_nhwc.shape[:3])
            and (x_nhwc.shape[0] * x_nhwc.shape[1] * x_nhwc.shape[2] + 64) * max(cin, cdy) * 4 < (1 << 31))


def conv1x1_wgrad(x_nhwc, dy_nhwc, cout, cin, pre_ab=None):
    """Weight gradient of a stride-1 1x1 conv -> dW [cout, cin, 1, 1] (the GEMM over positions of gemm1x1.hip).  ``pre_ab``: the
    conv's input was relu(pre_ab[0] x + pre_ab[1]) (BatchNorm + ReLU folded into its loader), x is the BatchNorm's input."""
    x, dy = _f32(x_nhwc), _f32(dy_nhwc)
    m = int(x.shape[0]) * int(x.shape[1]) * int(x.shape[2])
    nbytes = int(_hip.lib().dream_conv1x1_wgrad_workspace(m, cin, cout))
    ws = _workspace(nbytes, x.device)
    dw = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=x.device)
    if pre_ab is not None:
        call("dream_conv1x1_wgrad_pre_nhwc_f32", ptr(x), ptr(dy), ptr(dw), ptr(ws), m, cin, cout, int(dy.shape[3]), ptr(pre_ab), stream())
    else:
        call("dream_conv1x1_wgrad_nhwc_f32", ptr(x), ptr(dy), ptr(dw), ptr(ws), m, cin, cout, int(dy.shape[3]), stream())
    return dw


def conv_transpose4x4s2(x_nhwc, packed, cout, scale=None, shift=None, flags=0, direct_taps=16):
    """direct_taps: multiply-adds per (input pixel, cin, cout) of the direct algorithm this launch stands for -- 16 for a
    4x4 transposed conv, 36 when it replaces upsample + conv3x3; only used by bench.py's FLOP accounting."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    y = torch.empty((b, 2 * h, 2 * w, cout), dtype=torch.float32, device=x.device)
    call("dream_conv_transpose4x4s2_nhwc_f32", ptr(x), ptr(packed), ptr(scale), ptr(shift), ptr(y), b, h, w, cin, cout,
         int(packed.shape[-2]), flags, stream())
    return y


def bn_fold(bn_weight, bn_bias, running_mean, running_var, eps, conv_bias=None):
    c = int(bn_weight.numel())
    scale = torch.empty((c,), dtype=torch.float32, device=bn_weight.device)
    shift = torch.empty_like(scale)
    call("dream_bn_fold_f32", ptr(_f32(bn_weight)), ptr(_f32(bn_bias)), ptr(_f32(running_mean)), ptr(_f32(running_var)),
         ptr(conv_bias), float(eps), ptr(scale), ptr(shift), c, stream())
    return scale, shift


def im2col_nchw(x_nchw, kh, kw, stride, pad, kpad):
    x = _f32(x_nchw)
    b, c, h, w = (int(v) for v in x.shape)
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = torch.empty((b, ho, wo, kpad), dtype=torch.float32, device=x.device)
    call("dream_im2col_nchw_f32", ptr(x), ptr(y), b, c, h, w, kh, kw, stride, pad, kpad, stream())
    return y


def maxpool3s2(x_nhwc):
    x = _f32(x_nhwc)
    b, h, w, c = (int(v) for v in x.shape)
    y = torch.empty((b, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=torch.float32, device=x.device)
    call("dream_maxpool3s2_nhwc_f32", ptr(x), ptr(y), b, h, w, c, stream())
    return y


# ---- ResNet training operators -------------------------------------------------------------------------
def _workspace(nbytes, device):
    return torch.empty(((int(nbytes) + 7) // 8,), dtype=torch.float64, device=device)


def bn_train_fwd(x_nhwc, bn, residual=None, relu=True):
    """Train-mode BatchNorm2d on NHWC: returns (y, save_mean, save_invstd); updates bn's running stats."""
    x = _f32(x_nhwc)
    c = int(x.shape[-1])
    npix = x.numel() // c
    y = torch.empty_like(x)
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    ws = _workspace(_hip.lib().dream_bn_workspace(c), x.device)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    call("dream_bn_train_fwd_nhwc_f32", ptr(x), ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(residual), ptr(y),
         ptr(mean), ptr(invstd), ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked), ptr(ws), npix, c,
         float(bn.eps), float(momentum), 1 if relu else 0, stream())
    # the kernel wrote the running statistics through raw pointers: bump their version counters so that the folded
    # scale/shift of the evaluation path and a captured hipGraph (both keyed on _version) are rebuilt
    for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
        bump_version(t)
    return y, mean, invstd


_bump_log = threading.local()


class log_bumps:
    """Context manager: collects the tensors whose version is bumped inside it (a hipGraph capture of a training forward:
    the replay writes BatchNorm's running statistics again, so the owner of the graph repeats the bumps after every replay)."""

    def __enter__(self):
        self.prev = getattr(_bump_log, "items", None)
        _bump_log.items = []
        return _bump_log.items

    def __exit__(self, *exc):
        _bump_log.items = self.prev


def bump_version(t):
    """Tell autograd and the version-keyed caches that a kernel changed ``t`` in place through its raw pointer."""
    log = getattr(_bump_log, "items", None)
    if log is not None:
        log.append(t)
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is not None:
        inc(t)
    else:
        with torch.no_grad():
            t.add_(0)


def bn_train_bwd(x_nhwc, dy, y_act, gamma, mean, invstd, relu=True, want_g=False):
    """-> (dx, g or None, dgamma, dbeta)."""
    x = _f32(x_nhwc)
    c = int(x.shape[-1])
    npix = x.numel() // c
    dx = torch.empty_like(x)
    g = torch.empty_like(x) if want_g else None
    dgamma = torch.empty((c,), dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = _workspace(_hip.lib().dream_bn_workspace(c), x.device)
    call("dream_bn_train_bwd_nhwc_f32", ptr(x), ptr(_f32(dy)), ptr(y_act), ptr(gamma.detach()), ptr(mean), ptr(invstd), ptr(dx),
         ptr(g), ptr(dgamma), ptr(dbeta), ptr(ws), npix, c, 1 if relu else 0, stream())
    return dx, g, dgamma, dbeta


# ---- round 4: train-mode BatchNorm without its separate passes (csrc/bn.hip "round 4", csrc/gemm1x1.hip PRE / EPI) -----------------
def bn_counter_buffer(device, words=1 << 17):
    """Zero 32-bit words for the "last arriver finishes" tickets (every launch leaves its words zero again).  One buffer per
    model replica; each BatchNorm call site takes its own slice (bn_counter_slice), so launches on different streams never share."""
    return torch.zeros((int(words),), dtype=torch.int32, device=device)


def _ctr_words(counters, n):
    """``counters``: a zero int32 tensor of at least n words (tests), or an allocator n -> slice (a model's per-replica buffer)."""
    if callable(counters):
        out = counters(int(n))
        if out.numel() != n:
            raise RuntimeError("dream_amd: %d ticket words needed, the allocator returned %d" % (n, out.numel()))
        return out
    if counters.numel() < n:
        raise RuntimeError("dream_amd: %d ticket words needed, %d given" % (n, counters.numel()))
    return counters


# DREAM_BN_DEBUG=1 (round-4 advice): the "last arriver finishes" launches write their per-channel outputs from ONE wavefront that draws
# the last ticket; if a ticket word were ever left non-zero (an aborted launch) or the hand-ordered visibility failed, nobody would finish
# and the outputs -- torch.empty tensors -- would silently keep garbage.  The debug mode NaN-fills them before the launch, synchronises
# after it and raises unless every output is finite and every ticket word is zero again.  (Slow: one synchronisation per launch.)
BN_DEBUG = _os.environ.get("DREAM_BN_DEBUG", "0") == "1"


def _bn_debug_begin(*outs):
    if BN_DEBUG:
        for o in outs:
            o.fill_(float("nan"))


def _bn_debug_end(name, ctr, *outs):
    if BN_DEBUG:
        torch.cuda.synchronize(ctr.device) if ctr.is_cuda else None
        if int(ctr.abs().max()) != 0:
            raise RuntimeError("dream_amd (DREAM_BN_DEBUG): %s left non-zero ticket words" % name)
        for o in outs:
            if not bool(torch.isfinite(o).all()):
                raise RuntimeError("dream_amd (DREAM_BN_DEBUG): %s did not finish its per-channel outputs (the last-arriver wavefront never ran)" % name)


def _bn_args(bn):
    # nn.BatchNorm2d(momentum=None) means a cumulative moving average (factor 1 / num_batches_tracked), which the kernels' running-
    # statistics update does not implement; the reference's models use 0.1 (dream/models.py:45-47 and torchvision's default)
    if bn.momentum is None:
        raise NotImplementedError("dream_amd: BatchNorm2d(momentum=None) (cumulative moving average) is not supported by the HIP "
                                  "BatchNorm kernels; use a numeric momentum (the reference's models use 0.1)")
    momentum = bn.momentum
    return (ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
            float(bn.eps), float(momentum))


def _bn_outputs(c, device):
    ab = torch.empty((2, c), dtype=torch.float32, device=device)
    mean = torch.empty((c,), dtype=torch.float32, device=device)
    return ab, mean, torch.empty_like(mean)


def _bn_bump(bn):
    for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
        bump_version(t)


def bn_stats(z_nhwc, bn, counters):
    """Batch statistics of z in ONE launch -> (ab [2,C], save_mean, save_invstd); updates bn's running statistics."""
    z = _f32(z_nhwc)
    c = int(z.shape[-1])
    ab, mean, invstd = _bn_outputs(c, z.device)
    ws = _workspace(_hip.lib().dream_bn_stats_workspace(c), z.device)
    ctr = _ctr_words(counters, _hip.lib().dream_bn_stats_counters(c))
    _bn_debug_begin(ab, mean, invstd)
    call("dream_bn_stats_nhwc_f32", ptr(z), *_bn_args(bn), ptr(ab), ptr(mean), ptr(invstd), ptr(ws), ptr(ctr), z.numel() // c, c,
         stream())
    _bn_debug_end("bn_stats", ctr, ab, mean, invstd)
    _bn_bump(bn)
    return ab, mean, invstd


def bn_apply_ab(z_nhwc, ab, residual=None, relu=True):
    z = _f32(z_nhwc)
    c = int(z.shape[-1])
    y = torch.empty_like(z)
    call("dream_bn_apply_ab_nhwc_f32", ptr(z), ptr(ab), ptr(residual), ptr(y), z.numel() // c, c, 1 if relu else 0, stream())
    return y


def _mask_mode(y_act, ab):
    return 1 if y_act is not None else (2 if ab is not None else 0)


def bn_bwd_stats(z_nhwc, dy, mean, invstd, counters, y_act=None, ab=None):
    """-> (dgamma, dbeta) in ONE launch; the ReLU mask from ``y_act`` (> 0) or recomputed from (z, ab), none when both are None."""
    z = _f32(z_nhwc)
    c = int(z.shape[-1])
    dgamma = torch.empty((c,), dtype=torch.float32, device=z.device)
    dbeta = torch.empty_like(dgamma)
    ws = _workspace(_hip.lib().dream_bn_stats_workspace(c), z.device)
    ctr = _ctr_words(counters, _hip.lib().dream_bn_stats_counters(c))
    _bn_debug_begin(dgamma, dbeta)
    call("dream_bn_bwd_stats_nhwc_f32", ptr(z), ptr(_f32(dy)), ptr(y_act), ptr(ab), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta),
         ptr(ws), ptr(ctr), z.numel() // c, c, _mask_mode(y_act, ab), stream())
    _bn_debug_end("bn_bwd_stats", ctr, dgamma, dbeta)
    return dgamma, dbeta


def bn_bwd_apply(z_nhwc, dy, gamma, mean, invstd, dgamma, dbeta, y_act=None, ab=None, want_g=False):
    """-> (dz, g or None): dz = gamma invstd (g - dbeta / N - xhat dgamma / N), g = dy masked as in bn_bwd_stats."""
    z = _f32(z_nhwc)
    c = int(z.shape[-1])
    dz = torch.empty_like(z)
    g = torch.empty_like(z) if want_g else None
    call("dream_bn_bwd_apply_nhwc_f32", ptr(z), ptr(_f32(dy)), ptr(y_act), ptr(ab), ptr(gamma.detach()), ptr(mean), ptr(invstd),
         ptr(dgamma), ptr(dbeta), ptr(dz), ptr(g), z.numel() // c, c, _mask_mode(y_act, ab), stream())
    return dz, g


def conv1x1_bn(x_nhwc, packed, cout, bn, counters, pre_ab=None, shift=None):
    """[BatchNorm + ReLU of the producer, applied while loading: pre_ab ->] 1x1 conv -> (z, ab, save_mean, save_invstd) with the
    batch statistics of z summed in the GEMM's epilogue and finished inside the launch; updates bn's running statistics."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    m = b * h * w
    z = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    ab, mean, invstd = _bn_outputs(cout, x.device)
    ws = _workspace(_hip.lib().dream_conv1x1_bn_workspace(m, cout), x.device)
    ctr = _ctr_words(counters, _hip.lib().dream_conv1x1_bn_counters(m, cout))
    _bn_debug_begin(ab, mean, invstd)
    call("dream_conv1x1_bnstats_nhwc_f32", ptr(x), ptr(packed), ptr(shift), ptr(pre_ab), ptr(z), m, cin, cout, cin, *_bn_args(bn),
         ptr(ab), ptr(mean), ptr(invstd), ptr(ws), ptr(ctr), stream())
    _bn_debug_end("conv1x1_bn", ctr, ab, mean, invstd)
    _bn_bump(bn)
    return z, ab, mean, invstd


def conv1x1_bwd_bnmask(dy_nhwc, packed_t, cin, z_nhwc, ab, mean, invstd, counters, y_act=None, residual=None):
    """Data gradient of a 1x1 conv whose input was the output of a BN + ReLU: -> (g = (dy . w (+ residual)) masked, dgamma, dbeta of
    that BN).  The mask is recomputed from (z, ab) -- the input relu(BN(z)) was never stored -- or read from ``y_act`` (> 0)."""
    dy = _f32(dy_nhwc)
    b, h, w, k = (int(v) for v in dy.shape)
    m = b * h * w
    if tuple(z_nhwc.shape) != (b, h, w, cin):
        raise RuntimeError("conv1x1_bwd_bnmask: z shape %s != %s" % (tuple(z_nhwc.shape), (b, h, w, cin)))
    g = torch.empty((b, h, w, cin), dtype=torch.float32, device=dy.device)
    dgamma = torch.empty((cin,), dtype=torch.float32, device=dy.device)
    dbeta = torch.empty_like(dgamma)
    ws = _workspace(_hip.lib().dream_conv1x1_bn_workspace(m, cin), dy.device)
    ctr = _ctr_words(counters, _hip.lib().dream_conv1x1_bn_counters(m, cin))
    if residual is not None and tuple(residual.shape) != tuple(g.shape):
        raise RuntimeError("conv1x1_bwd_bnmask: residual shape %s != %s" % (tuple(residual.shape), tuple(g.shape)))
    _bn_debug_begin(dgamma, dbeta)
    call("dream_conv1x1_bwd_bnmask_nhwc_f32", ptr(dy), ptr(packed_t), ptr(residual), ptr(g), m, k, cin, k, ptr(_f32(z_nhwc)),
         None if y_act is not None else ptr(ab), ptr(y_act), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), ptr(ws), ptr(ctr), stream())
    _bn_debug_end("conv1x1_bwd_bnmask", ctr, dgamma, dbeta)
    return g, dgamma, dbeta


def conv3x3_winograd_bn(x_nhwc, packed_u, cout, bn, counters, shift=None):
    """Stride-1 3x3 conv on the Winograd F(2x2,3x3) kernel -> (z, ab, save_mean, save_invstd) with the batch statistics of z summed
    in the kernel's epilogue and finished inside the launch (cout a multiple of 64); updates bn's running statistics."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    lib = _hip.lib()
    z = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    ab, mean, invstd = _bn_outputs(cout, x.device)
    ws = _workspace(lib.dream_conv3x3_winograd_bn_workspace(b, h, w, cout), x.device)
    ctr = _ctr_words(counters, lib.dream_conv3x3_winograd_bn_counters(b, h, w, cout))
    _bn_debug_begin(ab, mean, invstd)
    call("dream_conv3x3_winograd_bnstats_nhwc_f32", ptr(x), ptr(packed_u), ptr(shift), ptr(z), b, h, w, cin, cout, *_bn_args(bn),
         ptr(ab), ptr(mean), ptr(invstd), ptr(ws), ptr(ctr), stream())
    _bn_debug_end("conv3x3_winograd_bn", ctr, ab, mean, invstd)
    _bn_bump(bn)
    return z, ab, mean, invstd


def conv3x3_winograd_bwd_bnmask(dy_nhwc, packed_u_t, cin, z_nhwc, ab, mean, invstd, counters):
    """Data gradient of a stride-1 3x3 conv whose input was relu(BN(z)): -> (g = conv3x3(dy; mode-1 weights) masked by
    [ab[0] z + ab[1] > 0], dgamma, dbeta of that BN), one launch of the Winograd F(2x2,3x3) kernel (cin a multiple of 64)."""
    dy = _f32(dy_nhwc)
    b, h, w, k = (int(v) for v in dy.shape)
    if tuple(z_nhwc.shape) != (b, h, w, cin):
        raise RuntimeError("conv3x3_winograd_bwd_bnmask: z shape %s != %s" % (tuple(z_nhwc.shape), (b, h, w, cin)))
    lib = _hip.lib()
    g = torch.empty((b, h, w, cin), dtype=torch.float32, device=dy.device)
    dgamma = torch.empty((cin,), dtype=torch.float32, device=dy.device)
    dbeta = torch.empty_like(dgamma)
    ws = _workspace(lib.dream_conv3x3_winograd_bn_workspace(b, h, w, cin), dy.device)
    ctr = _ctr_words(counters, lib.dream_conv3x3_winograd_bn_counters(b, h, w, cin))
    _bn_debug_begin(dgamma, dbeta)
    call("dream_conv3x3_winograd_bwd_bnmask_nhwc_f32", ptr(dy), ptr(packed_u_t), ptr(_f32(z_nhwc)), ptr(g), b, h, w, k, cin, ptr(ab),
         ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), ptr(ws), ptr(ctr), stream())
    _bn_debug_end("conv3x3_winograd_bwd_bnmask", ctr, dgamma, dbeta)
    return g, dgamma, dbeta


def channel_sum(x_nhwc):
    x = _f32(x_nhwc)
    c = int(x.shape[-1])
    out = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _workspace(_hip.lib().dream_bn_workspace(c), x.device)
    call("dream_channel_sum_nhwc_f32", ptr(x), ptr(out), ptr(ws), x.numel() // c, c, stream())
    return out


def conv2d_wgrad(x_nhwc, dy_nhwc, cout, cin, ksize, stride=1, flags=0, want_bias=False):
    """-> (dW [cout,cin,k,k], dbias [cout] or None).  x: [B,H,W,cin] (half-res with fused upsample), dy: [B,Ho,Wo,C>=cout]."""
    x, dy = _f32(x_nhwc), _f32(dy_nhwc)
    b, hs, ws_, _ = (int(v) for v in x.shape)
    scale = 2 if flags & CONV_UPSAMPLE2X else 1
    h, w = hs * scale, ws_ * scale
    cdy = int(dy.shape[3])
    if int(x.shape[3]) != cin:
        raise RuntimeError("wgrad: x has %d channels, expected %d" % (x.shape[3], cin))
    rows_pad = round_up(cdy, 64)
    nbytes = int(_hip.lib().dream_conv2d_wgrad_workspace(b, h, w, cin, rows_pad, ksize, stride))
    ws = _workspace(nbytes, x.device)
    nt = ksize * ksize
    dwp = torch.empty((nt, rows_pad, cin), dtype=torch.float32, device=x.device)
    dbias = torch.empty((cdy,), dtype=torch.float32, device=x.device) if want_bias else None
    call("dream_conv2d_wgrad_nhwc_f32", ptr(x), ptr(dy), ptr(dwp), ptr(dbias), ptr(ws), b, h, w, cin, cdy, rows_pad, ksize,
         stride, flags, stream())
    dw = torch.empty((cout, cin, ksize, ksize), dtype=torch.float32, device=x.device)
    call("dream_unpack_conv_weight", ptr(dwp), ptr(dw), cout, cin, nt, rows_pad, cin, stream())
    return dw, (dbias[:cout].contiguous() if want_bias else None)


def convT4x4_wgrad(x_nhwc, dy_nhwc):
    """ConvTranspose2d(4,2,1) weight gradient -> [Cin,Cout,4,4]."""
    x, dy = _f32(x_nhwc), _f32(dy_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    cout = int(dy.shape[3])
    rows_pad = round_up(cin, 64)
    ws = _workspace(_hip.lib().dream_convT4x4_wgrad_workspace(b, h, w, rows_pad, cout), x.device)
    dwp = torch.empty((16, rows_pad, cout), dtype=torch.float32, device=x.device)
    call("dream_convT4x4_wgrad_nhwc_f32", ptr(x), ptr(dy), ptr(dwp), ptr(ws), b, h, w, cin, rows_pad, cout, stream())
    dw = torch.empty((cin, cout, 4, 4), dtype=torch.float32, device=x.device)
    call("dream_unpack_conv_weight", ptr(dwp), ptr(dw), cin, cout, 16, rows_pad, cout, stream())
    return dw


def convT4x4_wgrad_winograd_applies(x_nhwc, dy_nhwc):
    """The nine-position F(2x2,2x2) weight gradient of ConvTranspose2d(4,2,1) takes channel counts that are multiples of 64 and a
    gradient tensor without channel padding at exactly twice the input's extent (DREAM_CONVT_WGRAD=direct: never)."""
    if _os.environ.get("DREAM_CONVT_WGRAD", "winograd") == "direct":
        return False
    b, h, w, cin = (int(v) for v in x_nhwc.shape)
    return (tuple(dy_nhwc.shape[:3]) == (b, 2 * h, 2 * w)
            and bool(_hip.lib().dream_convT4x4_wgrad_winograd_applies(cin, int(dy_nhwc.shape[3]))))


def convT4x4_wgrad_winograd(x_nhwc, dy_nhwc, want_bias=True):
    """ConvTranspose2d(4,2,1) weight gradient by minimal filtering (csrc/wgrad_wino.hip, CONVT) -> (dW [Cin,Cout,4,4], dbias [Cout] |
    None): 9 multiplications per 2x2 outputs of a phase instead of 16, the bias gradient from the same launch."""
    x, dy = _f32(x_nhwc), _f32(dy_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    cout = int(dy.shape[3])
    ws = _workspace(_hip.lib().dream_convT4x4_wgrad_winograd_workspace(b, h, w, cin, cout), x.device)
    dw = torch.empty((cin, cout, 4, 4), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_bias else None
    call("dream_convT4x4_wgrad_winograd_nhwc_f32", ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(ws), b, h, w, cin, cout, stream())
    return dw, db


def conv4x4s2(x_nhwc, packed16, cout, residual=None):
    """4x4 stride-2 pad-1 conv (data gradient of the 4x4 transposed conv)."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    y = torch.empty((b, h // 2, w // 2, cout), dtype=torch.float32, device=x.device)
    call("dream_conv4x4s2_nhwc_f32", ptr(x), ptr(packed16), ptr(residual), ptr(y), b, h, w, cin, cout,
         int(packed16.shape[-2]), 0, stream())
    return y


def pack_convT4x4_bwd_weight(wT):
    """[Cin_T,Cout_T,4,4] -> 16-tap packed [16][rows_pad >= Cin_T][Cout_T] for conv4x4s2."""
    w = _f32(wT)
    cin, cout = int(w.shape[0]), int(w.shape[1])
    rows_pad, cols_pad = _hip.cout_pad(cin), round_up(cout, 16)
    packed = torch.empty((16, rows_pad, cols_pad), dtype=torch.float32, device=w.device)
    call("dream_pack_conv_weight", ptr(w), ptr(packed), cin, cout, 16, rows_pad, cols_pad, 0, stream())
    return packed, cin


def conv2d_bwd_data(dy_nhwc, packed_t, cin, ksize, stride, in_hw, residual=None):
    """Data gradient of a k x k / stride conv: stride 1 -> same-size conv with mode-1 weights; stride 2 -> the transposed
    conv of the gradient (extent in_hw, which may be odd); with a residual to add, its zero-stuffed form."""
    dy = _f32(dy_nhwc)
    if stride == 1:
        return conv2d(dy, packed_t, cin, ksize, 1, None, None, residual, 0)
    b, ho, wo, c = (int(v) for v in dy.shape)
    h, w = in_hw
    y = torch.empty((b, h, w, cin), dtype=torch.float32, device=dy.device)
    if residual is None:          # transposed conv without the stuffed zeros (3x3: sub-pixel phases, 1x1: even positions)
        call("dream_conv2d_s2_bwd_data_nhwc_f32", ptr(dy), ptr(packed_t), ptr(y), b, ho, wo, c, h, w, cin,
             int(packed_t.shape[-2]), ksize, stream())
        return y
    call("dream_conv2d_nhwc_f32", ptr(dy), ptr(packed_t), None, None, ptr(residual), ptr(y), b, h, w, c, cin,
         int(packed_t.shape[-2]), ksize, 1, CONV_ZEROSTUFF2X, stream())
    return y


def maxpool3s2_idx(x_nhwc):
    """MaxPool2d(3,2,1) for training -> (y, idx): idx [B,Ho,Wo,C] uint8, the winner's position inside its 3x3 window."""
    x = _f32(x_nhwc)
    b, h, w, c = (int(v) for v in x.shape)
    y = torch.empty((b, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=torch.float32, device=x.device)
    idx = torch.empty(tuple(y.shape), dtype=torch.uint8, device=x.device)
    call("dream_maxpool3s2_idx_nhwc_f32", ptr(x), ptr(y), ptr(idx), b, h, w, c, stream())
    return y, idx


def maxpool3s2_idx_bwd(dy, idx, in_shape):
    """Backward of maxpool3s2_idx: dy [B,Ho,Wo,C], idx from the forward pass -> dx of ``in_shape`` = (B,H,W,C)."""
    b, h, w, c = (int(v) for v in in_shape)
    dx = torch.empty((b, h, w, c), dtype=torch.float32, device=dy.device)
    call("dream_maxpool3s2_idx_bwd_nhwc_f32", ptr(_f32(dy)), ptr(idx), ptr(dx), b, h, w, c, stream())
    return dx


def maxpool3s2_bwd(dy, x):
    b, h, w, c = (int(v) for v in x.shape)
    dx = torch.empty_like(x)
    call("dream_maxpool3s2_bwd_nhwc_f32", ptr(_f32(dy)), ptr(x), ptr(dx), b, h, w, c, stream())
    return dx


def clone(t):
    """``t.clone()`` by a kernel launch: a copy that stays a KERNEL node when the caller is being captured into a hipGraph (ATen's copy of
    a contiguous tensor is a hipMemcpyAsync -> a memcpy node; see dream_copy_f32 in include/dream_hip.h)."""
    src = _f32(t)
    out = torch.empty_like(src)
    call("dream_copy_f32", ptr(out), ptr(src), src.numel(), stream())
    return out


def add_(dst, src):
    call("dream_add_inplace_f32", ptr(dst), ptr(_f32(src)), dst.numel(), stream())
    return dst


def allreduce_sum_(flats, streams=None):
    """In place: every tensor of ``flats`` -- one flat fp32 buffer per data-parallel replica, each on its replica's GPU, same
    length -- becomes the element-wise sum of all of them (one RCCL all-reduce over xGMI for distinct GPUs, csrc/collective.hip).
    Ordered on ``streams[i]`` (a torch stream per buffer), default: the stream the calling thread currently uses on each device."""
    n = len(flats)
    count = int(flats[0].numel())
    for t in flats:
        if t.dtype != torch.float32 or int(t.numel()) != count:
            raise RuntimeError("allreduce_sum_: buffers must be float32 of equal length")
    devs = (ctypes.c_int * n)(*[(t.device.index or 0) if t.is_cuda else 0 for t in flats])
    bufs = (ctypes.c_void_p * n)(*[ptr(t) for t in flats])
    if streams is None:
        handles = [stream_on(t.device) if t.is_cuda else None for t in flats]
    else:
        handles = [None if st is None else st.cuda_stream for st in streams]
    call("dream_allreduce_sum_f32", n, devs, bufs, count, (ctypes.c_void_p * n)(*handles))
    return flats


def allreduce_uses_rccl(devices):
    n = len(devices)
    return bool(_hip.lib().dream_allreduce_uses_rccl(n, (ctypes.c_int * n)(*[int(d) for d in devices])))


def add(a, b, want_amax=False):
    """-> (a + b, amax or None)."""
    a, b = _f32(a), _f32(b)
    out = torch.empty_like(a)
    amax = new_amax(a.device) if want_amax else None
    call("dream_add_f32", ptr(a), ptr(b), ptr(out), a.numel(), ptr(amax), stream())
    return out, amax


def stage_input(img_nchw, maps_nchw, up, cpad, want_amax=False):
    """cat([image, nearest-upsample(maps, up)], dim=1) as zero-padded NHWC [B,H,W,cpad] (models.py:487-493)."""
    img, maps = _f32(img_nchw), _f32(maps_nchw)
    b, ci, h, w = (int(v) for v in img.shape)
    k = int(maps.shape[1])
    if (int(maps.shape[2]) * up, int(maps.shape[3]) * up) != (h, w) or int(maps.shape[0]) != b:
        raise RuntimeError("stage_input: maps %s x%d do not match image %s" % (tuple(maps.shape), up, tuple(img.shape)))
    out = torch.empty((b, h, w, cpad), dtype=torch.float32, device=img.device)
    amax = new_amax(img.device) if want_amax else None
    call("dream_stage_input_nhwc_f32", ptr(img), ptr(maps), ptr(out), b, h, w, ci, k, up, cpad, ptr(amax), stream())
    return out, amax


def stage_input_bwd(g_nhwc, ci, k, up):
    """Gradient of stage_input w.r.t. the maps: [B,H,W,C>=ci+k] -> [B,k,H/up,W/up] (sums over the up x up blocks)."""
    g = _f32(g_nhwc)
    b, h, w, c = (int(v) for v in g.shape)
    out = torch.empty((b, k, h // up, w // up), dtype=torch.float32, device=g.device)
    call("dream_stage_input_bwd_f32", ptr(g), ptr(out), b, h, w, ci, k, up, c, 0, stream())
    return out


# ---- split-precision (fp16x3) path ----------------------------------------------------------------------
def new_amax(device):
    """Device scalar (uint32 bit pattern of a non-negative float) that kernels atomicMax their max|y| into."""
    return torch.zeros((1,), dtype=torch.int32, device=device)


def absmax(x):
    a = new_amax(x.device)
    call("dream_absmax_f32", ptr(_f32(x)), x.numel(), ptr(a), stream())
    return a


def pack_conv_weight_f16x3(w_oihw, mode=0):
    """-> (hi, lo, exp, rows): two fp16 planes [ntaps][rows_pad][cols_pad] of w*2^exp and the device int exp."""
    w = _f32(w_oihw)
    cout, cin, kh, kw = (int(v) for v in w.shape)
    rows, cols = (cout, cin) if mode == 0 else (cin, cout)
    rows_pad, cols_pad = _hip.cout_pad(rows), round_up(cols, 32)
    hi = torch.empty((kh * kw, rows_pad, cols_pad), dtype=torch.float16, device=w.device)
    lo = torch.empty_like(hi)
    exp = torch.zeros((1,), dtype=torch.int32, device=w.device)
    scratch = torch.zeros((1,), dtype=torch.int32, device=w.device)
    call("dream_pack_conv_weight_f16x3", ptr(w), ptr(hi), ptr(lo), ptr(exp), ptr(scratch), cout, cin, kh * kw, rows_pad,
         cols_pad, mode, stream())
    return hi, lo, exp, rows


def conv2d_f16x3(x_nhwc, amax_in, packed16, cout, ksize, scale=None, shift=None, residual=None, flags=0, want_amax=True):
    """Split-precision conv (stride 1): returns (y, amax_out)."""
    hi, lo, exp, _ = packed16
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    if cin != hi.shape[-1]:
        raise RuntimeError("conv2d_f16x3: input has %d channels, packed weights expect %d" % (cin, hi.shape[-1]))
    if flags & (CONV_UPSAMPLE2X | CONV_ZEROSTUFF2X):
        h, w = 2 * h, 2 * w
    shape = (b, cout, h, w) if flags & CONV_OUT_NCHW else (b, h, w, cout)
    if flags & CONV_POOL2:
        shape = (b, h // 2, w // 2, cout)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    amax_out = new_amax(x.device) if want_amax else None
    call("dream_conv2d_f16x3_nhwc_f32", ptr(x), ptr(amax_in), ptr(hi), ptr(lo), ptr(exp), ptr(scale), ptr(shift),
         ptr(residual), ptr(y), ptr(amax_out), b, h, w, cin, cout, int(hi.shape[-2]), ksize, 1, flags, stream())
    return y, amax_out


def conv_transpose3x3s2_f16x3(x_nhwc, amax_in, packed16_mode1, cout, bias=None, relu=True):
    """Split-precision ConvTranspose2d(k3,s2,p1,output_padding 1) (+ReLU) by sub-pixel phases -> (y, amax_out)."""
    hi, lo, exp, _ = packed16_mode1
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    if cin != hi.shape[-1]:
        raise RuntimeError("conv_transpose3x3s2_f16x3: input has %d channels, packed weights expect %d" % (cin, hi.shape[-1]))
    y = torch.empty((b, 2 * h, 2 * w, cout), dtype=torch.float32, device=x.device)
    amax_out = new_amax(x.device)
    call("dream_conv_transpose3x3s2_f16x3_nhwc_f32", ptr(x), ptr(amax_in), ptr(hi), ptr(lo), ptr(exp), ptr(bias), ptr(y),
         ptr(amax_out), b, h, w, cin, cout, int(hi.shape[-2]), CONV_RELU if relu else 0, stream())
    return y, amax_out


def conv3x3_first_amax(x_nchw, w_oihw, bias, relu=True):
    x, w = _f32(x_nchw), _f32(w_oihw)
    b, cin, h, wd = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    y = torch.empty((b, h, wd, cout), dtype=torch.float32, device=x.device)
    amax = new_amax(x.device)
    call("dream_conv3x3_first_nchw_amax_f32", ptr(x), ptr(w), ptr(bias), ptr(y), ptr(amax), b, h, wd, cin, cout,
         1 if relu else 0, stream())
    return y, amax


def convT_wgrad(x_nhwc, dy_nhwc, ksize):
    """ConvTranspose2d(k=3 (output_padding 1) or 4, stride 2, pad 1) weight gradient -> [Cin,Cout,k,k]."""
    x, dy = _f32(x_nhwc), _f32(dy_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    cout = int(dy.shape[3])
    rows_pad = round_up(cin, 64)
    ws = _workspace(_hip.lib().dream_convT_wgrad_workspace(b, h, w, rows_pad, cout, ksize), x.device)
    nt = ksize * ksize
    dwp = torch.empty((nt, rows_pad, cout), dtype=torch.float32, device=x.device)
    call("dream_convT_wgrad_nhwc_f32", ptr(x), ptr(dy), ptr(dwp), ptr(ws), b, h, w, cin, rows_pad, cout, ksize, stream())
    dw = torch.empty((cin, cout, ksize, ksize), dtype=torch.float32, device=x.device)
    call("dream_unpack_conv_weight", ptr(dwp), ptr(dw), cin, cout, nt, rows_pad, cout, stream())
    return dw


def pack_convT4x4_weight_f16x3(wT):
    """ConvTranspose2d weight [Cin,Cout,4,4] -> (hi, lo, exp, cout): fp16 planes [4][4][rows_pad][cols_pad]."""
    w = _f32(wT)
    cin, cout = int(w.shape[0]), int(w.shape[1])
    rows_pad, cols_pad = _hip.cout_pad(cout), round_up(cin, 32)
    hi = torch.empty((4, 4, rows_pad, cols_pad), dtype=torch.float16, device=w.device)
    lo = torch.empty_like(hi)
    exp = torch.zeros((1,), dtype=torch.int32, device=w.device)
    scratch = torch.zeros((1,), dtype=torch.int32, device=w.device)
    call("dream_pack_convT4x4_weight_f16x3", ptr(w), ptr(hi), ptr(lo), ptr(exp), ptr(scratch), cin, cout, rows_pad, cols_pad,
         stream())
    return hi, lo, exp, cout


def conv_transpose4x4s2_f16x3(x_nhwc, amax_in, packed16, cout, scale=None, shift=None, flags=0, direct_taps=16):
    hi, lo, exp, _ = packed16
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    y = torch.empty((b, 2 * h, 2 * w, cout), dtype=torch.float32, device=x.device)
    amax_out = new_amax(x.device)
    call("dream_conv_transpose4x4s2_f16x3_nhwc_f32", ptr(x), ptr(amax_in), ptr(hi), ptr(lo), ptr(exp), ptr(scale), ptr(shift),
         ptr(y), ptr(amax_out), b, h, w, cin, cout, int(hi.shape[-2]), flags, stream())
    return y, amax_out


def conv2d_amax(x_nhwc, packed, cout, ksize, stride=1, scale=None, shift=None, residual=None, flags=0):
    """fp32 MFMA conv that also publishes max|y| (feeds the split-precision kernel's input scaling)."""
    x = _f32(x_nhwc)
    b, h, w, cin = (int(v) for v in x.shape)
    pad = ksize // 2
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    y = torch.empty((b, ho, wo, cout), dtype=torch.float32, device=x.device)
    amax = new_amax(x.device)
    call("dream_conv2d_amax_nhwc_f32", ptr(x), ptr(packed), ptr(scale), ptr(shift), ptr(residual), ptr(y), ptr(amax), b, h, w,
         cin, cout, int(packed.shape[-2]), ksize, stride, flags, stream())
    return y, amax
