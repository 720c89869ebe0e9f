"""Parity checks shared by the CPU suite (kernels run under the SIMT emulator, tests/emu) and the GPU
suite (-m gpu: the real libdream_hip.so through the C ABI).  Every check compares the HIP path with the
CPU oracle (oracle/, torch-CPU + NumPy) and/or the committed golden outputs of the real reference.

Tolerances: integer / index / peak-coordinate work is bit-exact; fp32 CNN values are compared with
TOL = 1e-4 * max(1, max|reference|) (BASELINE.json north_star: belief maps within 1e-4 in fp32 for maps of
O(1) magnitude; the synthetic recipe weights produce larger maps, hence the scale factor)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

import cases
from dream_amd import _hip, ops
import dream_amd
import dream_amd.optim
from oracle import models as om
from oracle import peaks as op

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def tol(ref):
    return TOL * max(1.0, float(np.abs(np.asarray(ref)).max()))


def to(dev, t):
    return t.to(dev) if dev != "cpu" else t


def check_conv(dev, B, H, W, Cin, Cout, flags, seed=0):
    g = torch.Generator().manual_seed(seed)
    ups = bool(flags & (ops.CONV_UPSAMPLE2X | ops.CONV_ZEROSTUFF2X))
    hs, ws = (H // 2, W // 2) if ups else (H, W)
    x = torch.randn(B, Cin, hs, ws, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    packed, rows, _, _ = ops.pack_weight(to(dev, w), 0)
    y = ops.conv3x3(to(dev, x.permute(0, 2, 3, 1).contiguous()), packed, to(dev, bias), Cout, flags).cpu()
    if flags & ops.CONV_UPSAMPLE2X:
        xr = F.interpolate(x, scale_factor=2)
    elif flags & ops.CONV_ZEROSTUFF2X:
        xr = torch.zeros(B, Cin, H, W)
        xr[:, :, ::2, ::2] = x
    else:
        xr = x
    ref = F.conv2d(xr, w, bias, padding=1)
    if flags & ops.CONV_RELU:
        ref = ref.relu()
    if flags & ops.CONV_POOL2:
        ref = F.max_pool2d(ref, 2)
    got = y if flags & ops.CONV_OUT_NCHW else y.permute(0, 3, 1, 2)
    err = float((got - ref).abs().max())
    assert err <= tol(ref.numpy()), (B, H, W, Cin, Cout, flags, err)
    return err


def check_conv_winograd(dev, B, H, W, Cin, Cout, flags=0, seed=0, mode=0, with_scale=False, residual=None, max_workgroups=(8, 24)):
    """Winograd F(2x2,3x3) conv vs an fp64 direct convolution: error relative to the output maximum at fp32 round-off level
    (<= 2e-6; measured ~3e-7, the direct fp32 kernel ~1.5e-7)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5 if with_scale else None
    if mode == 1:                                        # data-gradient operator of conv(w'): w' is [Cout_x, Cin_x] = [Cin, Cout] here
        wsrc = torch.randn(Cin, Cout, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        w = wsrc.permute(1, 0, 2, 3).flip(2, 3).contiguous()
        packed, rows = ops.pack_weight_winograd(to(dev, wsrc), 1)
    else:
        packed, rows = ops.pack_weight_winograd(to(dev, w), 0)
    assert rows == Cout
    res = None
    if residual is not None:
        res = torch.randn(B, Cout, H, W, generator=g)
    args = (to(dev, _nhwc(x)), packed, Cout, to(dev, scale) if with_scale else None, to(dev, bias),
            to(dev, _nhwc(res)) if res is not None else None, flags)
    y = ops.conv3x3_winograd(*args).cpu().permute(0, 3, 1, 2)
    # the persistent grid sized for fewer resident workgroups (each walks over several tile blocks, prefetching the next
    # block's first chunk during the last chunk of the current one): same kernels, same order of operations per output
    for cap in max_workgroups:
        _hip.lib().dream_conv3x3_winograd_set_max_workgroups(cap)
        try:
            y_cap = ops.conv3x3_winograd(*args).cpu().permute(0, 3, 1, 2)
        finally:
            _hip.lib().dream_conv3x3_winograd_set_max_workgroups(0)
        assert torch.equal(y, y_cap), ("persistent grid", cap, float((y - y_cap).abs().max()))
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    if with_scale:
        ref = ref * scale.double().view(1, -1, 1, 1)
    ref = ref + bias.double().view(1, -1, 1, 1)
    if res is not None:
        ref = torch.where(res.double() > 0, ref, torch.zeros_like(ref)) if flags & ops.CONV_RELUMASK else ref + res.double()
    if flags & ops.CONV_RELU:
        ref = ref.relu()
    if flags & ops.CONV_POOL2:
        ref = F.max_pool2d(ref, 2)
    assert y.shape == ref.shape, (tuple(y.shape), tuple(ref.shape))
    err = float((y.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    assert err <= 2e-6, (B, H, W, Cin, Cout, flags, err)
    return err


def check_conv_winograd4_pool_both(dev, B, H, W, Cin, Cout, seed=0):
    """csrc/conv_wino4.hip MODE 4 (the training forward of a conv that feeds MaxPool2d(2)): the un-pooled and the pooled tensor from ONE
    launch equal the plain launch and a max-pool over its output bit for bit (odd extents: the pool floors, as nn.MaxPool2d)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = ((torch.rand(Cout, Cin, 3, 3, generator=g) * 2 - 1) * (6.0 / (9 * Cin)) ** 0.5).to(dev)
    bias = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    u, rows = ops.pack_weight_winograd4(w, 0)
    y_ref = ops.conv3x3_winograd4(x, u, rows, None, bias, None, ops.CONV_RELU)
    p_ref = ops.maxpool2(y_ref)
    p_fused = ops.conv3x3_winograd4(x, u, rows, None, bias, None, ops.CONV_RELU | ops.CONV_POOL2)
    y, p = ops.conv3x3_winograd4_pool_both(x, u, rows, bias, ops.CONV_RELU)
    assert tuple(p.shape) == (B, H // 2, W // 2, Cout) and tuple(y.shape) == (B, H, W, Cout)
    assert torch.equal(y, y_ref), float((y - y_ref).abs().max())
    assert torch.equal(p, p_ref) and torch.equal(p, p_fused)
    return 0.0


def check_conv_winograd4(dev, B, H, W, Cin, Cout, flags=0, seed=0, mode=0, with_scale=False, residual=None, max_workgroups=(8,), tol=1e-5):
    """Winograd F(4x4,3x3) conv (csrc/conv_wino4.hip) vs an fp64 direct convolution: error relative to the output maximum
    <= 1e-5 (measured 1e-6 .. 6.5e-6 on these unit-variance inputs with the interpolation points 0, +-1, 1/2, -2 -- 512 input
    channels the largest; F(2x2,3x3) ~3e-7 .. 8e-7, the direct fp32 kernel ~1.5e-7).  The persistent grid capped to a few workgroups must give the same bits."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5 if with_scale else None
    if mode == 1:
        wsrc = torch.randn(Cin, Cout, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
        w = wsrc.permute(1, 0, 2, 3).flip(2, 3).contiguous()
        packed, rows = ops.pack_weight_winograd4(to(dev, wsrc), 1)
    else:
        packed, rows = ops.pack_weight_winograd4(to(dev, w), 0)
    assert rows == Cout
    res = torch.randn(B, Cout, H, W, generator=g) if residual is not None else None
    args = (to(dev, _nhwc(x)), packed, Cout, to(dev, scale) if with_scale else None, to(dev, bias),
            to(dev, _nhwc(res)) if res is not None else None, flags)
    y = ops.conv3x3_winograd4(*args).cpu().permute(0, 3, 1, 2)
    for cap in max_workgroups:
        _hip.lib().dream_conv3x3_winograd4_set_max_workgroups(cap)
        try:
            y_cap = ops.conv3x3_winograd4(*args).cpu().permute(0, 3, 1, 2)
        finally:
            _hip.lib().dream_conv3x3_winograd4_set_max_workgroups(0)
        assert torch.equal(y, y_cap), ("persistent grid", cap, float((y - y_cap).abs().max()))
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    if with_scale:
        ref = ref * scale.double().view(1, -1, 1, 1)
    ref = ref + bias.double().view(1, -1, 1, 1)
    if res is not None:
        ref = torch.where(res.double() > 0, ref, torch.zeros_like(ref)) if flags & ops.CONV_RELUMASK else ref + res.double()
    if flags & ops.CONV_RELU:
        ref = ref.relu()
    if flags & ops.CONV_POOL2:
        ref = F.max_pool2d(ref, 2)
    assert y.shape == ref.shape, (tuple(y.shape), tuple(ref.shape))
    err = float((y.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    assert err <= tol, (B, H, W, Cin, Cout, flags, err)
    return err


def check_convT4x4_winograd(dev, B, H, W, Cin, Cout, flags=0, seed=0, with_scale=False, max_workgroups=(8,), tile=2):
    """ConvTranspose2d(k4,s2,p1) by minimal filtering on the Winograd kernels (tile 2: F(2x2,3x3) with the 9-position phase patterns,
    tile 4: F(4x4,3x3) with the 25-position ones) vs an fp64 conv_transpose2d: error relative to the output maximum at fp32 round-off
    level (tile 4: the F(4x4) class, <= 1e-5)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    wT = torch.randn(Cin, Cout, 4, 4, generator=g) * (2.0 / (4 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5 if with_scale else None
    u4, rows = ops.pack_convT4x4_winograd_weight_tile(to(dev, wT), tile)
    assert rows == Cout
    ref = F.conv_transpose2d(x.double(), wT.double(), None, stride=2, padding=1)
    if with_scale:
        ref = ref * scale.double().view(1, -1, 1, 1)
    ref = ref + bias.double().view(1, -1, 1, 1)
    if flags & ops.CONV_RELU:
        ref = ref.relu()
    args = (tile, to(dev, _nhwc(x)), u4, Cout, to(dev, scale) if with_scale else None, to(dev, bias), flags)
    y = ops.conv_transpose4x4s2_winograd_tile(*args).cpu().permute(0, 3, 1, 2)
    assert y.shape == ref.shape, (tuple(y.shape), tuple(ref.shape))
    err = float((y.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    # F(4x4): the error grows with the square root of the reduction length -- 6e-6 at 512 input channels, 1.5e-5 at 2048 (a shape
    # ops.convT4x4_winograd_tile never gives to this kernel: its 13x13 map has too few tiles)
    assert err <= ((2e-5 if Cin > 1024 else 1e-5) if tile == 4 else 2e-6), (B, H, W, Cin, Cout, flags, tile, err)
    setter = _hip.lib().dream_conv3x3_winograd4_set_max_workgroups if tile == 4 else _hip.lib().dream_conv3x3_winograd_set_max_workgroups
    for cap in max_workgroups:                          # persistent grid sized for fewer workgroups: same result, bit for bit
        setter(cap)
        try:
            y_cap = ops.conv_transpose4x4s2_winograd_tile(*args).cpu().permute(0, 3, 1, 2)
        finally:
            setter(0)
        assert torch.equal(y, y_cap), ("persistent grid", cap)
    return err


def check_conv4x4s2_winograd(dev, B, H, W, Cin, Cout, seed=0, tile=2):
    """Data gradient of ConvTranspose2d(k4,s2,p1) (x [B,Cin,H,W] -> y [B,Cout,2H,2W]) by the four phase convs on the Winograd
    kernels (tile 2: nine positions each; tile 4: 25 of 36) vs fp64 autograd: error relative to the maximum of the result at fp32
    round-off level (tile 4: the F(4x4) class)."""
    g = torch.Generator().manual_seed(seed)
    wT = torch.randn(Cin, Cout, 4, 4, generator=g) * (2.0 / (4 * Cout)) ** 0.5
    dy = torch.randn(B, Cout, 2 * H, 2 * W, generator=g)
    xref = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(xref, wT.double(), None, stride=2, padding=1).backward(dy.double())
    u4, rows = ops.pack_convT4x4_winograd_weight_tile(to(dev, wT), tile, 1)
    assert rows == Cin
    dx = ops.conv4x4s2_winograd_tile(tile, to(dev, _nhwc(dy)), u4, Cin).cpu().permute(0, 3, 1, 2)
    assert dx.shape == xref.grad.shape
    err = float((dx.double() - xref.grad).abs().max()) / max(1.0, float(xref.grad.abs().max()))
    assert err <= (1e-5 if tile == 4 else 2e-6), (B, H, W, Cin, Cout, tile, err)
    return err


def check_conv1x1(dev, B, H, W, Cin, Cout, flags=0, seed=0, mode=0, with_scale=False, residual=False, ksplits=(0, 1, 2, 4)):
    """LDS-free GEMM kernel for stride-1 1x1 convs vs an fp64 evaluation: error relative to the output maximum at fp32 round-off
    level, for every K split the shape admits (the split only changes the order of the sum)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    bias = torch.randn(Cout, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5 if with_scale else None
    if mode == 1:                                        # data-gradient operator of conv(w'): w' is [Cin, Cout] here
        wsrc = torch.randn(Cin, Cout, 1, 1, generator=g) * (2.0 / Cin) ** 0.5
        w = wsrc.permute(1, 0, 2, 3).contiguous()
        packed, rows = ops.pack_conv1x1_weight(to(dev, wsrc), 1)
    else:
        w = torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5
        packed, rows = ops.pack_conv1x1_weight(to(dev, w), 0)
    assert rows == Cout
    res = torch.randn(B, Cout, H, W, generator=g) if residual else None
    ref = F.conv2d(x.double(), w.double(), None)
    if with_scale:
        ref = ref * scale.double().view(1, -1, 1, 1)
    ref = ref + bias.double().view(1, -1, 1, 1)
    if res is not None:
        ref = ref + res.double()
    if flags & ops.CONV_RELU:
        ref = ref.relu()
    worst = 0.0
    for ks in ksplits:
        if ks and Cin % (32 * ks):
            continue
        first = None
        for rows_forced in (0, 64, 32):                  # wavefront tile height: by problem size, 64 rows, 32 rows (round 6)
            _hip.lib().dream_conv1x1_set_ksplit(ks)
            _hip.lib().dream_conv1x1_set_rows(rows_forced)
            try:
                y = ops.conv1x1(to(dev, _nhwc(x)), packed, Cout, to(dev, scale) if with_scale else None, to(dev, bias),
                                to(dev, _nhwc(res)) if res is not None else None, flags).cpu().permute(0, 3, 1, 2)
            finally:
                _hip.lib().dream_conv1x1_set_ksplit(0)
                _hip.lib().dream_conv1x1_set_rows(0)
            assert y.shape == ref.shape, (tuple(y.shape), tuple(ref.shape))
            err = float((y.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
            assert err <= 2e-6, (B, H, W, Cin, Cout, flags, ks, rows_forced, err)
            worst = max(worst, err)
            if ks:                                         # a forced K split: the same sums in the same order whatever the tile height
                first = y if first is None else first
                assert torch.equal(y, first), (B, H, W, Cin, Cout, ks, rows_forced)
    return worst


def check_conv1x1_wgrad(dev, B, H, W, Cin, Cout, seed=0, pad_dy=0):
    """GEMM weight gradient of the stride-1 1x1 conv vs an fp64 evaluation: error relative to sum |terms| at fp32 round-off level."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, H, W, Cout, generator=g)
    ref = torch.einsum("bhwo,bhwi->oi", dy.double(), x.double())
    aref = torch.einsum("bhwo,bhwi->oi", dy.double().abs(), x.double().abs())
    dyn = torch.cat([dy, torch.randn(B, H, W, pad_dy, generator=g)], dim=3).contiguous() if pad_dy else dy
    dw = ops.conv1x1_wgrad(to(dev, x), to(dev, dyn), Cout, Cin)
    assert tuple(dw.shape) == (Cout, Cin, 1, 1)
    err = float(((dw.cpu().double().view(Cout, Cin) - ref).abs() / aref).max())
    assert err <= 3e-6, (B, H, W, Cin, Cout, err)
    return err


def check_wgrad_winograd(dev, B, H, W, Cin, Cout, seed=0, pad_dy=0, ups=False):
    """Winograd-domain weight gradient vs an fp64 evaluation of the direct sums: error relative to sum |terms| at fp32
    round-off level (the direct MFMA kernel is checked the same way).  ups: the conv ran on the nearest x2 upsample of x
    (x is [B,Cin,H/2,W/2]); the upsample is fused into the kernel's patch loads."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H // 2, W // 2, generator=g) if ups else torch.randn(B, Cin, H, W, generator=g)
    xl = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    dy = torch.randn(B, Cout, H, W, generator=g)
    wref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xl.double(), wref, None, padding=1).backward(dy.double())
    aref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)      # sum |terms|: the error scale
    F.conv2d(xl.double().abs(), aref, None, padding=1).backward(dy.double().abs())
    dyn = _nhwc(dy)
    if pad_dy:
        dyn = torch.cat([dyn, torch.zeros(B, H, W, pad_dy)], dim=3).contiguous()
    dw, db = ops.conv3x3_wgrad_winograd(to(dev, _nhwc(x)), to(dev, dyn), Cout, Cin, flags=ops.CONV_UPSAMPLE2X if ups else 0)
    err = float(((dw.cpu().double() - wref.grad).abs() / aref.grad).max())
    assert dw.shape == (Cout, Cin, 3, 3) and err <= 3e-6, (B, H, W, Cin, Cout, err)
    assert float((db.cpu().double() - dy.double().sum((0, 2, 3))).abs().max()) <= 1e-5 * float(dy.abs().sum((0, 2, 3)).max())
    if ops.WGRAD_BIAS_FUSION and Cin % 64 == 0 and Cout % 64 == 0:
        # the bias gradient rode in the kernel's dy loader: the weight gradient must not notice (same bits as without it), padded dy
        # channels (garbage here) must not leak into it, and it agrees with the stand-alone column sums
        junk = torch.cat([_nhwc(dy), torch.full((B, H, W, 16), 1e6)], dim=3).contiguous()
        dwj, dbj = ops.conv3x3_wgrad_winograd(to(dev, _nhwc(x)), to(dev, junk), Cout, Cin, flags=ops.CONV_UPSAMPLE2X if ups else 0)
        ops.WGRAD_BIAS_FUSION = False
        try:
            dw2, db2 = ops.conv3x3_wgrad_winograd(to(dev, _nhwc(x)), to(dev, dyn), Cout, Cin, flags=ops.CONV_UPSAMPLE2X if ups else 0)
        finally:
            ops.WGRAD_BIAS_FUSION = True
        if ups and ops.UPS_WGRAD_AS_CONVT:
            # round 6: an upsample + conv's gradient with un-padded dy runs on the nine-position transposed-conv form; the padded-dy and
            # the no-bias launches above took the sixteen-position kernel with the fused upsample: two correct fp32 evaluations
            scale = float(aref.grad.max())
            assert float((dw - dwj).abs().max()) <= 3e-6 * scale and float((dw - dw2).abs().max()) <= 3e-6 * scale
            assert float((db - dbj).abs().max()) <= 1e-5 * float(dy.abs().sum((0, 2, 3)).max())
        else:
            assert torch.equal(dw, dw2) and torch.equal(dw, dwj) and torch.equal(db, dbj)
        assert float((db - db2).abs().max()) <= 1e-5 * float(dy.abs().sum((0, 2, 3)).max())
    return err


def check_convT4x4_wgrad_winograd(dev, B, H, W, Cin, Cout, seed=0):
    """nn.ConvTranspose2d(k4, s2, p1) weight gradient on the nine-position F(2x2,2x2) form (csrc/wgrad_wino.hip CONVT) against an
    fp64 evaluation of torch's conv_transpose2d backward: error relative to sum |terms| at fp32 round-off level, every one of the
    sixteen taps written (the four phases' 2 x 2 taps tile the 4 x 4 kernel), the bias gradient from the same launch, and agreement
    with the direct kernel it replaces."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, 2 * H, 2 * W, generator=g)
    wref = torch.zeros(Cin, Cout, 4, 4, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(x.double(), wref, None, stride=2, padding=1).backward(dy.double())
    aref = torch.zeros(Cin, Cout, 4, 4, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(x.double().abs(), aref, None, stride=2, padding=1).backward(dy.double().abs())
    xd, dyd = to(dev, _nhwc(x)), to(dev, _nhwc(dy))
    assert ops.convT4x4_wgrad_winograd_applies(xd, dyd)
    dw = torch.full((Cin, Cout, 4, 4), float("nan"))
    dw, db = ops.convT4x4_wgrad_winograd(xd, dyd)
    assert dw.shape == (Cin, Cout, 4, 4) and bool(torch.isfinite(dw).all())
    err = float(((dw.cpu().double() - wref.grad).abs() / aref.grad).max())
    assert err <= 3e-6, (B, H, W, Cin, Cout, err)
    assert float((db.cpu().double() - dy.double().sum((0, 2, 3))).abs().max()) <= 1e-5 * float(dy.abs().sum((0, 2, 3)).max())
    dw2, db2 = ops.convT4x4_wgrad_winograd(xd, dyd, want_bias=False)
    assert db2 is None and torch.equal(dw, dw2)               # the bias sums ride along without touching the weight gradient
    direct = ops.convT4x4_wgrad(xd, dyd)
    assert float((dw - direct).abs().max()) <= 3e-6 * float(aref.grad.max())
    return err


def check_conv_transpose(dev, B, H, W, Cin, Cout, seed=0):
    """ConvTranspose2d(k3,s2,p1,op1) == zero-stuffed conv with mode-1 packed weights."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    wt = torch.randn(Cin, Cout, 3, 3, generator=g) * (2.0 / (2.25 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    packed, rows, _, _ = ops.pack_weight(to(dev, wt), 1)
    assert rows == Cout
    y = ops.conv3x3(to(dev, x.permute(0, 2, 3, 1).contiguous()), packed, to(dev, bias), Cout,
                    ops.CONV_ZEROSTUFF2X).cpu()
    ref = F.conv_transpose2d(x, wt, bias, stride=2, padding=1, output_padding=1)
    err = float((y.permute(0, 3, 1, 2) - ref).abs().max())
    assert err <= tol(ref.numpy()), err
    return err


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def check_conv2d_general(dev, B, H, W, Cin, Cout, k, stride, seed=0):
    """k x k / stride conv with the fused BN-scale, shift, residual and ReLU epilogue (ResNet Bottleneck)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, None, stride=stride, padding=k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g)
    ref = (ref + res).relu()
    packed, rows, _ = ops.pack_conv_weight(to(dev, w), 0)
    y = ops.conv2d(to(dev, _nhwc(x)), packed, Cout, k, stride, to(dev, scale), to(dev, shift), to(dev, _nhwc(res)),
                   ops.CONV_RELU).cpu()
    err = float((y.permute(0, 3, 1, 2) - ref).abs().max())
    assert err <= tol(ref.numpy()), (B, H, W, Cin, Cout, k, stride, err)


def check_conv_transpose4x4(dev, B, H, W, Cin, Cout, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    wT = torch.randn(Cin, Cout, 4, 4, generator=g) * (2.0 / (4 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x, wT, bias, stride=2, padding=1).relu()
    packed, cout = ops.pack_convT4x4_weight(to(dev, wT))
    y = ops.conv_transpose4x4s2(to(dev, _nhwc(x)), packed, cout, None, to(dev, bias), ops.CONV_RELU).cpu()
    err = float((y.permute(0, 3, 1, 2) - ref).abs().max())
    assert err <= tol(ref.numpy()), err


def check_resnet_stem(dev, B, H, W):
    x = torch.randn(B, 3, H, W)
    w = torch.randn(64, 3, 7, 7) * 0.1
    ref = F.conv2d(x, w, None, stride=2, padding=3)
    col = ops.im2col_nchw(to(dev, x), 7, 7, 2, 3, 160)
    # the patch matrix itself, bit for bit: k = (c * 7 + ky) * 7 + kx as F.unfold orders it, zeros beyond k = 147 and outside the image
    ho, wo = ref.shape[2], ref.shape[3]
    unf = F.unfold(x, 7, padding=3, stride=2).reshape(B, 147, ho, wo).permute(0, 2, 3, 1)
    assert torch.equal(col.cpu()[..., :147], unf) and float(col.cpu()[..., 147:].abs().max()) == 0.0
    packed, rows, _ = ops.pack_matrix_weight(to(dev, w.reshape(64, 147)), 160)
    y = ops.conv2d(col, packed, 64, 1, 1)
    assert float((y.cpu().permute(0, 3, 1, 2) - ref).abs().max()) <= tol(ref.numpy())
    mp = ops.maxpool3s2(y).cpu().permute(0, 3, 1, 2)
    assert torch.equal(mp, F.max_pool2d(y.cpu().permute(0, 3, 1, 2), 3, 2, 1))
    gam, bet, mean, var = torch.rand(64) + 0.5, torch.randn(64), torch.randn(64), torch.rand(64) + 0.5
    sc, sh = ops.bn_fold(to(dev, gam), to(dev, bet), to(dev, mean), to(dev, var), 1e-5)
    bn = F.batch_norm(ref, mean, var, gam, bet, False, 0.1, 1e-5)
    assert float((ref * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1) - bn).abs().max()) <= tol(bn.numpy())


def check_first_conv(dev, B, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(64, generator=g)
    y = ops.conv3x3_first(to(dev, x), to(dev, w), to(dev, b), relu=True).cpu()
    ref = F.conv2d(x, w, b, padding=1).relu()
    assert float((y.permute(0, 3, 1, 2) - ref).abs().max()) <= tol(ref.numpy())


def check_pool_and_layouts(dev):
    x = torch.randn(2, 9, 14, 8)
    y = ops.maxpool2(to(dev, x)).cpu()
    assert torch.equal(y, F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))
    x = torch.randn(2, 7, 13, 17)
    assert torch.equal(ops.nchw_to_nhwc(to(dev, x)).cpu(), x.permute(0, 2, 3, 1))
    p = ops.nchw_to_nhwc(to(dev, x), 16).cpu()
    assert torch.equal(p[..., :7], x.permute(0, 2, 3, 1)) and float(p[..., 7:].abs().max()) == 0.0
    assert torch.equal(ops.nhwc_to_nchw(to(dev, x.permute(0, 2, 3, 1).contiguous())).cpu(), x)
    # the pixels a stride-2 1x1 conv reads, gathered, and the transpose (round 6: the trunk's downsample convs on the 1x1 GEMM)
    for shape in ((2, 9, 14, 8), (1, 25, 25, 64), (3, 1, 6, 4), (2, 8, 1, 12)):
        x = torch.randn(*shape)
        ys = ops.subsample2(to(dev, x))
        assert torch.equal(ys.cpu(), x[:, ::2, ::2, :])
        full = torch.zeros_like(x)
        full[:, ::2, ::2, :] = x[:, ::2, ::2, :]
        assert torch.equal(ops.scatter2(ys, shape[1], shape[2]).cpu(), full)
    # the patch rows of a 3x3 stride-2 pad-1 conv and their transpose (round 6: ResNet's layer4.0.conv2 on the 1x1 GEMM)
    for shape in ((2, 9, 14, 8), (1, 25, 25, 16), (2, 4, 6, 4), (1, 1, 5, 4)):
        x = torch.randn(*shape)
        b, h, w_, c = shape
        unf = F.unfold(x.permute(0, 3, 1, 2), 3, padding=1, stride=2)                  # [B, C * 9, L], channel-major
        ho, wo = (h - 1) // 2 + 1, (w_ - 1) // 2 + 1
        ref = unf.reshape(b, c, 9, ho, wo).permute(0, 3, 4, 2, 1).reshape(b, ho, wo, 9 * c)
        col = ops.im2col3s2(to(dev, x))
        assert torch.equal(col.cpu(), ref)
        g = torch.randn(b, ho, wo, 9 * c)
        back = F.fold(g.reshape(b, ho, wo, 9, c).permute(0, 4, 3, 1, 2).reshape(b, c * 9, ho * wo), (h, w_), 3, padding=1, stride=2)
        got = ops.col2im3s2(to(dev, g), h, w_).cpu()
        assert float((got - back.permute(0, 2, 3, 1)).abs().max()) <= 1e-5
    # ConvTranspose2d(k4, s2, p1) = per-pixel tap contributions (a 1x1 GEMM) + col2im4s2 (round 6: the decoder's first layer on small maps)
    for (b, h, w_, cin, cout) in ((2, 5, 7, 6, 8), (1, 13, 13, 4, 4), (1, 1, 1, 3, 4)):
        x, wt, bias = torch.randn(b, h, w_, cin), torch.randn(cin, cout, 4, 4), torch.randn(cout)
        g = torch.einsum("bhwi,iokl->bhwklo", x.double(), wt.double()).reshape(b, h, w_, 16 * cout).float()
        ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), stride=2, padding=1).permute(0, 2, 3, 1)
        got = ops.col2im4s2(to(dev, g), cout, to(dev, bias)).cpu()
        assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        sc = torch.rand(cout) + 0.5
        got = ops.col2im4s2(to(dev, g), cout, to(dev, bias), scale=to(dev, sc), flags=ops.CONV_RELU).cpu()
        ref2 = ((ref - bias.double()) * sc.double() + bias.double()).clamp_min(0.0)
        assert float((got.double() - ref2).abs().max()) <= 1e-5 * max(1.0, float(ref2.abs().max()))
    w = torch.randn(40, 24, 3, 3)
    packed, rows, rp, cp = ops.pack_weight(to(dev, w), 0)
    assert rows == 40 and rp % 128 == 0 and cp == 32
    pk = packed.cpu()
    assert torch.equal(pk[:, :40, :24], w.permute(2, 3, 0, 1).reshape(9, 40, 24))
    assert float(pk[:, 40:].abs().max()) == 0.0 and float(pk[:, :, 24:].abs().max()) == 0.0


def peak_golden():
    return np.load(os.path.join(GOLD, "peaks_golden.npz"))


def check_peaks_case(dev, name):
    """Bit-exact: smoothed map == scipy restatement, peak list == reference output, keypoints == reference."""
    g = peak_golden()
    maps, off = cases.peak_cases()[name]
    m = to(dev, torch.from_numpy(maps))
    sm = ops.gaussian_sigma3(m).cpu().numpy()
    for i in range(len(maps)):
        assert np.array_equal(sm[i], op.gaussian_filter_sigma3(maps[i])), (name, i)
    kps, counts = ops.keypoints_from_belief_maps(m[None], off)
    assert np.array_equal(counts.cpu().numpy()[0], g[name + "/counts"])
    assert np.array_equal(kps.cpu().numpy(), g[name + "/keypoints"])
    xy, sc, cn = (t.cpu().numpy() for t in ops.peaks_list(m, off, cap=4))
    flat_xy = [xy[i, :cn[i]] for i in range(len(maps))]
    flat_sc = [sc[i, :cn[i]] for i in range(len(maps))]
    assert np.array_equal(np.concatenate(flat_xy).reshape(-1, 2), g[name + "/xy"].reshape(-1, 2))
    assert np.array_equal(np.concatenate(flat_sc), g[name + "/score"])


def check_peak_rule_settings(dev):
    """use_belief_peak_scores / belief_peak_next_best_score (dream/network.py:189-191) reach the kernel: keypoints equal the
    reference's for every setting of tests/golden/peak_rule_golden.npz, through ops and through DreamNetwork.inference."""
    g = np.load(os.path.join(GOLD, "peak_rule_golden.npz"))
    net = build_network("vgg_q", dev)
    net.enable_evaluation()
    for name in cases.PEAK_RULE_CASES:
        maps, off = cases.peak_cases()[name]
        m = to(dev, torch.from_numpy(maps))[None]
        changed = 0
        for tag, (use, thr) in cases.PEAK_RULE_SETTINGS.items():
            kps, _ = ops.keypoints_from_belief_maps(m, off, use, thr)
            assert np.array_equal(kps.cpu().numpy(), g[name + "/" + tag]), (name, tag)
            assert np.array_equal(op.keypoints_from_belief_maps(maps[None], off, use, thr), g[name + "/" + tag]), (name, tag)
            changed += int(not np.array_equal(g[name + "/" + tag], peak_golden()[name + "/keypoints"]))
        if name == "two_blobs_gap":
            assert changed >= 3                                         # the settings do change the outcome
            net.model = lambda x, _m=m: [_m]                              # as make_golden.py drives the reference
            net.network_config["training"]["config"]["net_output_resolution"] = [100, 100]
            for tag, (use, thr) in cases.PEAK_RULE_SETTINGS.items():
                net.use_belief_peak_scores, net.belief_peak_next_best_score = use, thr
                assert np.array_equal(net.inference(m)[1].numpy(), g[name + "/" + tag]), tag


def check_peaks_api(dev):
    """dream_amd.image_proc.peaks_from_belief_maps reproduces the reference's own KAT
    (test/test_image_proc.py:94-120) and its return structure."""
    maps = torch.from_numpy(op.create_belief_map((80, 60), [np.array([65.0, 20.0]), np.array([100.0, 80.0])])).float()
    all_peaks = dream_amd.image_proc.peaks_from_belief_maps(to(dev, maps), 0.0)
    assert len(all_peaks[0]) == 1 and len(all_peaks[1]) == 0
    assert np.linalg.norm(np.array([65.0, 20.0]) - np.array(all_peaks[0][0][:2])) < 1.0e-3
    assert all_peaks[0][0][3] == 0


def check_softargmax(dev):
    g = np.load(os.path.join(GOLD, "softargmax_golden.npz"))
    for name, (maps, beta) in cases.softargmax_cases().items():
        out = ops.softargmax(to(dev, torch.from_numpy(maps)), to(dev, torch.ones(maps.shape[1]) * beta)).cpu().numpy()
        assert np.abs(out - g[name]).max() <= 1e-4 * max(1.0, np.abs(g[name]).max()), name


def build_network(arch, dev, weights=None, optimizer="adam", lr=1e-4, in_res=None, quiet=True):
    import contextlib
    import io
    if arch in om.VARIANTS:
        base, over = om.VARIANTS[arch]
        k, manip = 7, "panda"
        cfg = dream_amd.default_network_config(base, manip, optimizer=optimizer, learning_rate=lr)
        cfg["architecture"].update(over)
    else:
        k = cases.CNN_CASES[arch][0]
        manip = cases.CNN_CASES[arch][1]
        cfg = dream_amd.default_network_config(arch, manip, optimizer=optimizer, learning_rate=lr)
    if in_res is not None:
        cfg["training"]["config"]["net_input_resolution"] = list(in_res)
    with contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext():
        net = dream_amd.create_network_from_config_data(cfg)
    if weights is None:
        weights = om.recipe_weights(om.build_model(arch, k).state_dict())
    net.model.load_state_dict({"module." + key: v for key, v in weights.items()})
    return net


def check_conv_f16x3(dev, B, H, W, Cin, Cout, k, flags, x_scale=1.0, w_scale=0.1, seed=0):
    """Split-precision conv vs an fp64 reference: error relative to the output maximum must stay in the fp32 class
    (<= 5e-6; torch's own fp32 conv is ~5e-7 on these shapes), including with an outlier that sets the tensor scale
    and with extreme tensor magnitudes; the published amax must equal max|y|."""
    g = torch.Generator().manual_seed(seed)
    ups = bool(flags & ops.CONV_UPSAMPLE2X)
    x = torch.randn(B, Cin, H // 2 if ups else H, W // 2 if ups else W, generator=g) * x_scale
    x[0, 0, 0, 0] = 40 * x_scale
    w = torch.randn(Cout, Cin, k, k, generator=g) * w_scale
    bias = torch.randn(Cout, generator=g) * x_scale * w_scale
    xr = F.interpolate(x, scale_factor=2) if ups else x
    ref = F.conv2d(xr.double(), w.double(), bias.double(), padding=k // 2)
    if flags & ops.CONV_RELU:
        ref = ref.relu()
    if flags & ops.CONV_POOL2:
        ref = F.max_pool2d(ref, 2)
    p16 = ops.pack_conv_weight_f16x3(to(dev, w), 0)
    amax_in = ops.absmax(to(dev, x))
    y, amax_out = ops.conv2d_f16x3(to(dev, _nhwc(x)), amax_in, p16, Cout, k, None, to(dev, bias), None, flags)
    got = y.cpu() if flags & ops.CONV_OUT_NCHW else y.cpu().permute(0, 3, 1, 2)
    scale = float(ref.abs().max())
    err = float((got.double() - ref).abs().max()) / scale
    assert err <= 5e-6, (B, H, W, Cin, Cout, k, flags, err)
    am = float(np.frombuffer(amax_out.cpu().numpy().tobytes(), dtype=np.float32)[0])
    assert abs(am - float(got.abs().max())) <= 1e-6 * scale
    return err


def check_model_inference(dev, arch, shape, precision="fp32"):
    """belief maps within TOL of the reference's golden output; detections agree; coordinates within
    1e-3 px (they are bit-exact functions of maps that differ in the last bits)."""
    b, h, w = shape
    g = np.load(os.path.join(GOLD, "cnn_%s.npz" % arch))
    net = build_network(arch, dev)
    net.enable_evaluation()
    if precision != "fp32":
        net.model.module.precision = precision
    x = torch.from_numpy(cases.image_batch(b, h, w, seed=b * 1000 + h))
    with torch.no_grad():
        maps, kps = net.inference(to(dev, x))
    assert kps.device.type == "cpu" and kps.dtype == torch.float32
    tag = "%dx%dx%d" % (b, h, w)
    y = maps.cpu().numpy()
    if tag + "/maps" in g:
        ref = g[tag + "/maps"]
        err = np.abs(y - ref).max()
    else:
        ref = g[tag + "/maps_sample"]
        err = np.abs(y[:, :, ::7, ::7] - ref).max()
    assert err <= tol(ref), (arch, tag, err)
    ref_k = g[tag + "/keypoints"]
    got_k = kps.numpy()
    # the peak stage itself is bit-exact on the maps the HIP CNN produced
    off = op.upsampling_offset(*net.trained_net_output_resolution())
    assert np.array_equal(got_k, op.keypoints_from_belief_maps(y, off))
    same = (got_k == np.float32(-999.999)) == (ref_k == np.float32(-999.999))
    assert same.mean() >= 0.9          # a borderline 0.25-rule / threshold decision may flip on 1e-5 map noise
    both = (got_k != np.float32(-999.999)) & (ref_k != np.float32(-999.999))
    assert np.abs(got_k - ref_k)[both].max(initial=0.0) < 0.5
    return float(err)


def check_structured(dev, arch, precision="fp32", repeat=1):
    """North-star bounds on the structured fixture (blob-like maps of magnitude 1, tests/golden/structured_<arch>.npz,
    generated by the reference): belief maps within an ABSOLUTE 1e-4, every detection / rejection decision identical,
    detected keypoints within 1e-3 px of the reference's.  ``repeat``: the fixture's frames tiled to a batch of repeat x its size
    (the BASELINE batch sizes: the algorithm selection and the grids of the benchmarked configuration), EVERY copy held to the bounds."""
    case = arch
    arch, manip, k, last, (b, h, w), recipe, zero_bg = cases.STRUCTURED_CASES[case]
    g = np.load(os.path.join(GOLD, "structured_%s.npz" % case))
    sd = om.build_model(arch, k).state_dict()
    weights = {"structured": om.structured_weights, "smooth": om.smooth_weights, "recipe": om.recipe_weights}[recipe](sd)
    weights[last + ".weight"] = torch.from_numpy(g["final_weight"])
    weights[last + ".bias"] = torch.from_numpy(g["final_bias"])
    net = build_network(arch, dev, weights=weights, in_res=(w, h))
    net.enable_evaluation()
    if precision != "fp32":
        net.model.module.precision = precision
    x, _ = cases.structured_input(case)
    with torch.no_grad():
        maps, kps = net.inference(to(dev, torch.from_numpy(x).repeat(repeat, 1, 1, 1)))
    y, got_k, ref_k = maps.cpu().numpy(), kps.numpy(), g["keypoints"]
    ref_maps = g["maps"]
    if repeat > 1:
        assert y.shape[0] == repeat * b and got_k.shape[0] == repeat * b
        ref_maps = np.concatenate([ref_maps] * repeat, axis=0)
        ref_k = np.concatenate([ref_k] * repeat, axis=0)
    err = float(np.abs(y - ref_maps).max())
    assert float(g["maps"].max()) <= 1.0 + 1e-6 and float(np.abs(g["maps"]).max()) <= (8.0 if recipe == "smooth" else 1.0 + 1e-6)
    assert err <= 1e-4, (arch, precision, err)
    det = ref_k[..., 0] > -999
    assert np.array_equal(got_k[..., 0] > -999, det), "detection decisions differ from the reference"
    if cases.STRUCTURED_MIN_DETECTIONS.get(case, 0.25) > 0:
        assert 0 < det.sum() < det.size
    else:
        assert det.sum() < det.size                        # (a case kept for its map values: every keypoint may be rejected)
    perr = float(np.abs(got_k - ref_k)[det].max()) if det.any() else 0.0
    assert perr <= 1e-3, (arch, precision, perr)
    assert np.array_equal(got_k[~det], ref_k[~det])                 # the -999.999 sentinels, bit for bit
    return err, perr


def check_train_steps(dev, opt, steps=3):
    """DreamNetwork.train() against the reference's golden losses / grad norms / updated parameters."""
    g = np.load(os.path.join(GOLD, "train_vgg_q_%s.npz" % opt))
    w = om.recipe_weights(om.build_model("vgg_q", 7).state_dict(), cases.TRAIN_FINAL_KEYS, cases.TRAIN_FINAL_SCALE)
    net = build_network("vgg_q", dev, weights=w, optimizer=opt, lr=cases.TRAIN_LR[opt], in_res=(96, 64))
    net.enable_training()
    x = to(dev, torch.from_numpy(cases.image_batch(2, 64, 96, seed=5)))
    t = to(dev, torch.from_numpy(cases.target_batch(2, 7, (24, 16), in_wh=(96, 64), seed=5)))
    losses = []
    for step in range(steps):
        loss = net.train([x], t)
        losses.append(loss.item())
        if step == 0:
            for key, p in net.model.named_parameters():
                ref = float(g["gradnorm/" + key])
                assert abs(float(p.grad.double().norm()) - ref) <= 1e-3 * max(ref, 1e-9), key
    # The first loss is a pure forward pass: 1e-4 whatever the optimizer.  Later losses under Adam follow parameters whose update
    # is lr x sign-like for elements with round-off-level gradients (see below): measured 1.0e-4 on the third loss with F(4x4,3x3)
    # on every layer it serves, 1e-6 with F(2x2): held to 5e-4 ONLY while the tests force F(4x4) onto this 2-frame batch
    # (ops.set_winograd_tile(4)); by the layer-wise rule (F(2x2) at this size) and under SGD: 1e-4 throughout.
    assert np.allclose(losses[:1], g["losses"][:1], rtol=1e-4), (losses, g["losses"])
    forced4 = ops._WINOGRAD_TILE_FORCED == 4
    assert np.allclose(losses, g["losses"][:steps], rtol=5e-4 if (opt == "adam" and forced4) else 1e-4), (losses, g["losses"], forced4)
    if steps == 3:
        # Updated parameters.  SGD: the update is lr x gradient, held tight.  Adam: the update is lr x m / (sqrt(v) + eps) --
        # a sign flip of a gradient element at round-off level (two correct fp32 convolution algorithms differ there) moves
        # the parameter by up to 2 lr per step; so at least 98 % of the sampled values must agree tightly and none may be
        # further off than what such flips can produce.
        close, total, worst = 0, 0, 0.0
        for key, p in net.model.named_parameters():
            s = p.detach().flatten()[:: max(1, p.numel() // 64)][:64].cpu().numpy()
            ref = g["param_sample/" + key]
            ok = np.isclose(s, ref, rtol=1e-3, atol=2e-6)
            if opt != "adam":
                assert ok.all(), key
            close, total, worst = close + int(ok.sum()), total + ok.size, max(worst, float(np.abs(s - ref).max()))
        assert close >= 0.98 * total and worst <= 2.0 * steps * cases.TRAIN_LR[opt] + 2e-6, (close, total, worst)
        return close / total, worst


def check_backward_ops(dev):
    for (B, H, W, Cin, Cout, ups) in [(2, 9, 11, 32, 64, 0), (1, 8, 12, 64, 48, 0), (1, 8, 12, 32, 16, 1)]:
        x = torch.randn(B, Cin, H // 2 if ups else H, W // 2 if ups else W, requires_grad=True)
        w = (torch.randn(Cout, Cin, 3, 3) * 0.1).requires_grad_()
        bias = torch.zeros(Cout, requires_grad=True)
        y = F.conv2d(F.interpolate(x, scale_factor=2) if ups else x, w, bias, padding=1)
        dy = torch.randn_like(y)
        y.backward(dy)
        xh = to(dev, x.detach().permute(0, 2, 3, 1).contiguous())
        dyh = to(dev, dy.permute(0, 2, 3, 1).contiguous())
        dw, db = ops.conv3x3_wgrad(xh, dyh, Cout, Cin, ops.CONV_UPSAMPLE2X if ups else 0)
        packed_t, rows, _, _ = ops.pack_weight(to(dev, w.detach()), 1)
        dx = ops.conv3x3(dyh, packed_t, None, rows, 0)
        if ups:
            dx = ops.upsample2_bwd(dx)
        assert float((dw.cpu() - w.grad).abs().max()) <= tol(w.grad.numpy())
        assert float((db.cpu() - bias.grad).abs().max()) <= tol(bias.grad.numpy())
        assert float((dx.cpu().permute(0, 3, 1, 2) - x.grad).abs().max()) <= tol(x.grad.numpy())
    x = torch.randn(2, 3, 18, 21)
    w = (torch.randn(64, 3, 3, 3) * 0.1).requires_grad_()
    b = torch.zeros(64, requires_grad=True)
    y = F.conv2d(x, w, b, padding=1)
    dy = torch.randn_like(y)
    y.backward(dy)
    dw, db = ops.conv3x3_first_wgrad(to(dev, x), to(dev, dy.permute(0, 2, 3, 1).contiguous()))
    assert float((dw.cpu() - w.grad).abs().max()) <= tol(w.grad.numpy())
    assert float((db.cpu() - b.grad).abs().max()) <= tol(b.grad.numpy())
    x = torch.randn(2, 8, 9, 10, requires_grad=True)
    y = F.max_pool2d(x, 2)
    dy = torch.randn_like(y)
    y.backward(dy)
    dx = ops.maxpool2_bwd(to(dev, dy.permute(0, 2, 3, 1).contiguous()), to(dev, x.detach().permute(0, 2, 3, 1).contiguous()))
    assert torch.equal(dx.cpu().permute(0, 3, 1, 2), x.grad)
    for shape, ties in (((2, 8, 9, 11), False), ((1, 4, 7, 7), True), ((3, 12, 6, 5), True), ((1, 4, 2, 3), False)):   # odd extents: leftover row / column zeroed
        x = (torch.round(torch.randn(*shape) * 2) if ties else torch.randn(*shape)).requires_grad_()
        for relu in (False, True):
            x.grad = None
            y = F.max_pool2d(x.relu() if relu else x, 2)
            dy = torch.randn_like(y)
            y.backward(dy)
            src = x.detach().relu() if relu else x.detach()
            dx = ops.maxpool2_bwd(to(dev, _nhwc(dy)), to(dev, _nhwc(src)), relu=relu)
            if relu and ties:
                # at x == 0 torch's ReLU gradient is 0, and so is ours (best > 0); elsewhere identical
                pass
            assert torch.equal(dx.cpu().permute(0, 3, 1, 2), x.grad), (shape, ties, relu)
    # MaxPool2d(2) o ReLU backward in one pass, and the ReLU mask applied in a data-gradient conv's epilogue
    x = torch.randn(2, 8, 9, 10, requires_grad=True)
    y = F.max_pool2d(x.relu(), 2)
    dy = torch.randn_like(y)
    y.backward(dy)
    dx = ops.maxpool2_bwd(to(dev, _nhwc(dy)), to(dev, _nhwc(x.detach().relu())), relu=True)
    assert torch.equal(dx.cpu().permute(0, 3, 1, 2), x.grad)
    x = torch.randn(2, 32, 7, 9, requires_grad=True)
    w = torch.randn(48, 32, 3, 3) * 0.1
    y = F.conv2d(x.relu(), w, padding=1)
    dy = torch.randn_like(y)
    y.backward(dy)
    packed_t, rows, _, _ = ops.pack_weight(to(dev, w), 1)
    dx = ops.conv3x3(to(dev, _nhwc(dy)), packed_t, None, rows, 0, relu_mask=to(dev, _nhwc(x.detach().relu())))
    assert float((dx.cpu().permute(0, 3, 1, 2) - x.grad).abs().max()) <= tol(x.grad.numpy())
    assert torch.equal(dx.cpu().permute(0, 3, 1, 2) == 0, x.grad == 0)
    o, t = torch.randn(2, 7, 5, 6), torch.randn(2, 7, 5, 6)
    l, gr = ops.mse_fwd_bwd(to(dev, o), to(dev, t))
    assert abs(l.item() - F.mse_loss(o, t).item()) < 1e-6
    assert float((gr.cpu() - 2 * (o - t) / o.numel()).abs().max()) < 1e-7
    o2 = (o * 2).requires_grad_()
    lh = F.smooth_l1_loss(o2, t)
    lh.backward()
    l, gr = ops.mse_fwd_bwd(to(dev, o2.detach()), to(dev, t), kind="huber")
    assert abs(l.item() - lh.item()) < 1e-6 and float((gr.cpu() - o2.grad).abs().max()) < 1e-7
    crit = dream_amd.optim.HipSmoothL1Loss()
    o3 = to(dev, (o * 2)).requires_grad_()
    crit(o3, to(dev, t)).backward()
    assert float((o3.grad.cpu() - o2.grad).abs().max()) < 1e-7


def check_resnet_train_step(dev, arch="resnet_h", shape=(2, 64, 64), steps=1):
    """One DreamNetwork.train() step of a ResNet (train-mode BatchNorm + full backward).  Train-mode BN over the
    few samples of a small test batch is ill-conditioned (two correct fp32 implementations differ by percents in
    the deep layers), so the yardstick is the fp64 oracle: the HIP path must be as close to it as torch's own fp32
    CPU path is (factor 3 + a 1e-4 floor), on the loss, on every parameter gradient and on the BN buffers."""
    k = cases.CNN_CASES[arch][0]
    b, h, w = shape
    wts = om.recipe_weights(om.build_model(arch, k).state_dict())
    ref32 = om.build_model(arch, k)
    ref32.load_state_dict(wts)
    ref32.train()
    ref64 = om.build_model(arch, k)
    ref64.load_state_dict(wts)
    ref64.double().train()
    net = build_network(arch, dev, weights=wts, optimizer="sgd", lr=0.0, in_res=(w, h))
    net.enable_training()
    out_wh = net.net_output_resolution_from_input_resolution((w, h))
    x = torch.from_numpy(cases.image_batch(b, h, w, seed=3))
    t = torch.from_numpy(cases.target_batch(b, k, out_wh, in_wh=(w, h), seed=3))
    l32 = F.mse_loss(ref32(x)[0], t)
    l32.backward()
    l64 = F.mse_loss(ref64(x.double())[0], t.double())
    l64.backward()
    loss = net.train([to(dev, x)], to(dev, t))
    assert abs(loss.item() - l64.item()) <= 3 * abs(l32.item() - l64.item()) + 1e-4 * abs(l64.item())
    # direction of every parameter gradient vs the fp64 truth: a wiring error (missing branch, wrong tap, wrong
    # BN term) drops the cosine far below what fp32 round-off + ReLU-mask flips cost torch's own fp32 path
    def cos(a, b_):
        return float((a * b_).sum() / (a.norm() * b_.norm() + 1e-300))
    gnorm = max(p.grad.norm().item() for p in ref64.parameters())
    c32, chip = [], []
    for (name, p32), (_, p64), (_, pm) in zip(ref32.named_parameters(), ref64.named_parameters(),
                                              net.model.module.named_parameters()):
        if p64.grad.norm().item() < 1e-7 * gnorm:       # conv biases in front of a BN: true gradient is exactly 0
            assert pm.grad.detach().cpu().double().norm().item() < 1e-4 * gnorm, name
            continue
        c32.append(cos(p32.grad.double(), p64.grad))
        chip.append((cos(pm.grad.detach().cpu().double(), p64.grad), name))
    floor = min(0.98, min(c32) - 0.02)
    bad = [(n, c) for c, n in chip if c < floor]
    assert not bad, (floor, bad[:5])
    for (name, b32), (_, b64), (_, bm) in zip(ref32.named_buffers(), ref64.named_buffers(), net.model.module.named_buffers()):
        if name.startswith("_") or "beta_const" in name:
            continue
        scale = b64.double().abs().max().item() + 1e-6
        e32 = (b32.double() - b64.double()).abs().max().item() / scale
        ehip = (bm.detach().cpu().double() - b64.double()).abs().max().item() / scale
        assert ehip <= 3 * e32 + 1e-3, (name, ehip, e32)


def grad_sample(t, n=64):
    """The sampling rule of tests/golden/make_golden.py (G12)."""
    f = t.detach().flatten()
    return f[:: max(1, f.numel() // n)][:n].double().cpu().numpy().copy()


def check_resnet_train_golden(dev, case):
    """One DreamNetwork.train() step of a ResNet against the REFERENCE's own step (tests/golden/train_<case>.npz, generated by
    make_golden.py G12 from dream/network.py:328-364 on dream/models.py:17-155).  Loss to 1e-4 (measured 1e-6).  The decoder -- 4 (5) x
    [ConvTranspose2d, BatchNorm, ReLU] + the 1x1 head -- is held to 1 % on every gradient norm, 0.995 on the direction of the sampled
    gradients, 1e-6 on the updated parameters, 5e-4 on its running statistics.  The trunk's gradients pass through ~100 train-mode BatchNorms over 2 frames (ill-conditioned: two correct fp32
    implementations differ by percents) and are held by direction: cosine of the sampled gradients >= 0.98 overall and per
    stage, norms within 5 %.  -> dict of the measured figures."""
    arch, manip, k, (b, h, w), final_keys = cases.RESNET_TRAIN_CASES[case]
    g = np.load(os.path.join(GOLD, "train_%s.npz" % case))
    wts = om.recipe_weights(om.build_model(arch, k).state_dict(), final_keys, cases.TRAIN_FINAL_SCALE)
    net = build_network(arch, dev, weights=wts, optimizer="sgd", lr=cases.RESNET_TRAIN_LR, in_res=(w, h))
    net.enable_training()
    ow, oh = net.trained_net_output_resolution()
    x = to(dev, torch.from_numpy(cases.image_batch(b, h, w, seed=17)))
    t = to(dev, torch.from_numpy(cases.target_batch(b, k, (ow, oh), in_wh=(w, h), seed=17)))
    loss = net.train([x], t).item()
    ref_loss = float(g["loss"])
    res = {"loss_rel": abs(loss - ref_loss) / abs(ref_loss)}
    assert res["loss_rel"] <= 1e-4, (loss, ref_loss)
    dec_norm, dec_param = 0.0, 0.0
    dec = [0.0, 0.0, 0.0]
    trunk = {}
    for key, p in net.model.named_parameters():
        name = key[len("module."):]
        ref_n, ref_s = float(g["gradnorm/" + key]), g["gradsample/" + key]
        got_n, got_s = float(p.grad.double().norm()), grad_sample(p.grad)
        if name.startswith(cases.RESNET_DECODER_PREFIXES):
            wn = float(g["gradnorm/module." + name.replace(".bias", ".weight")])
            if name.endswith(".bias") and ref_n <= 1e-5 * wn:     # bias of a transposed conv in front of a BatchNorm: the true
                assert got_n <= 1e-4 * wn, (key, got_n, wn)      # gradient is exactly zero, what both sides hold is round-off
                continue
            dec_norm = max(dec_norm, abs(got_n - ref_n) / ref_n)
            dec[0] += float((got_s * ref_s).sum()) / (ref_n * ref_n)          # per-tensor normalised: every tensor counts alike
            dec[1] += float((got_s * got_s).sum()) / (ref_n * ref_n)
            dec[2] += float((ref_s * ref_s).sum()) / (ref_n * ref_n)
            ps, rs_ = grad_sample(p), g["param_sample/" + key]
            dec_param = max(dec_param, float(np.abs(ps - rs_).max()) / max(float(np.abs(rs_).max()), 1e-30))
        else:
            stage = name.split(".")[0]
            acc = trunk.setdefault(stage, [0.0, 0.0, 0.0, 0.0, 0.0])
            acc[0] += float((got_s * ref_s).sum())
            acc[1] += float((got_s * got_s).sum())
            acc[2] += float((ref_s * ref_s).sum())
            acc[3] += got_n ** 2
            acc[4] += ref_n ** 2
    res.update(decoder_gradnorm_rel=dec_norm, decoder_grad_cos=dec[0] / (dec[1] * dec[2]) ** 0.5, decoder_param_rel=dec_param)
    # the decoder's gradients inherit the trunk's round-off through its input (the trunk output differs in the 3rd digit between
    # two correct fp32 implementations at 4x4 / 2x2 maps and 2 frames): norms to 1 %, direction to 0.995, updated parameters exact
    assert dec_norm <= 1e-2 and res["decoder_grad_cos"] >= 0.995 and dec_param <= 1e-6, res
    tot = [sum(a[i] for a in trunk.values()) for i in range(5)]
    res["trunk_cos"] = tot[0] / (tot[1] * tot[2]) ** 0.5
    res["trunk_norm_rel"] = abs(tot[3] ** 0.5 - tot[4] ** 0.5) / tot[4] ** 0.5
    res["trunk_stage_cos"] = {st: a[0] / max((a[1] * a[2]) ** 0.5, 1e-300) for st, a in trunk.items() if a[2] > 0}
    assert res["trunk_cos"] >= 0.98 and res["trunk_norm_rel"] <= 0.05, res
    assert min(res["trunk_stage_cos"].values()) >= 0.95, res
    bn_dec, bn_trunk = 0.0, 0.0
    for key, buf in net.model.named_buffers():
        if not (key.endswith("running_mean") or key.endswith("running_var")):
            continue
        ref_s, got_s = g["buffer_sample/" + key], grad_sample(buf)
        e = float(np.abs(got_s - ref_s).max()) / max(float(np.abs(ref_s).max()), 1e-30)
        if key[len("module."):].startswith(cases.RESNET_DECODER_PREFIXES):
            bn_dec = max(bn_dec, e)
        else:
            bn_trunk = max(bn_trunk, e)
    res.update(decoder_bn_rel=bn_dec, trunk_bn_rel=bn_trunk)
    assert bn_dec <= 5e-4 and bn_trunk <= 2e-3, res      # measured 3e-5 .. 1.2e-4 / 9e-5 .. 3.6e-4
    return res


def check_resnet_training_ops(dev):
    def nchw(t):
        return t.permute(0, 3, 1, 2)
    for (B, C, H, W, relu, res) in [(2, 64, 5, 7, True, True), (3, 256, 4, 4, True, False), (2, 2048, 3, 3, False, False),
                                    (1, 48, 6, 5, True, False)]:
        bn = torch.nn.BatchNorm2d(C)
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_()
        bn2 = torch.nn.BatchNorm2d(C)
        bn2.load_state_dict(bn.state_dict())
        bn2 = bn2.to(dev) if dev != "cpu" else bn2
        x = torch.randn(B, C, H, W, requires_grad=True)
        r = torch.randn(B, C, H, W) if res else None
        y_ref = bn(x)
        if res:
            y_ref = y_ref + r
        if relu:
            y_ref = y_ref.relu()
        dy = torch.randn_like(y_ref)
        y_ref.backward(dy)
        versions = [t._version for t in (bn2.running_mean, bn2.running_var, bn2.num_batches_tracked)]
        y, mean, invstd = ops.bn_train_fwd(to(dev, _nhwc(x.detach())), bn2, to(dev, _nhwc(r)) if res else None, relu)
        # the version-keyed caches (folded eval-mode scale/shift, captured graphs) must see the updated statistics
        assert all(t._version > v for t, v in zip((bn2.running_mean, bn2.running_var, bn2.num_batches_tracked), versions))
        dx, g, dgam, dbet = ops.bn_train_bwd(to(dev, _nhwc(x.detach())), to(dev, _nhwc(dy)), y, bn2.weight, mean, invstd, relu,
                                             want_g=True)
        assert float((nchw(y.cpu()) - y_ref).abs().max()) < 1e-5
        assert float((bn.running_mean - bn2.running_mean.cpu()).abs().max()) < 1e-6
        assert float((bn.running_var - bn2.running_var.cpu()).abs().max()) < 1e-5
        assert int(bn2.num_batches_tracked.item()) == 1
        assert float((nchw(dx.cpu()) - x.grad).abs().max()) < 1e-5
        assert float((dgam.cpu() - bn.weight.grad).abs().max()) < 1e-4 and float((dbet.cpu() - bn.bias.grad).abs().max()) < 1e-4
        gm = dy * (y_ref > 0) if relu else dy
        assert float((nchw(g.cpu()) - gm).abs().max()) == 0.0
    # (.., 128/192 x 128/160 1x1) -> the 128x128-tile blocking, with a ragged last row / column block and a bias
    for (B, H, W, Cin, Cout, k, s_) in [(2, 9, 11, 64, 48, 1, 1), (1, 13, 13, 32, 128, 1, 2), (2, 12, 10, 32, 64, 3, 2),
                                        (1, 13, 25, 64, 32, 3, 2), (2, 9, 11, 128, 192, 1, 1), (1, 13, 13, 160, 128, 1, 2)]:
        x = torch.randn(B, Cin, H, W, requires_grad=True)
        w = (torch.randn(Cout, Cin, k, k) * 0.1).requires_grad_()
        y = F.conv2d(x, w, None, stride=s_, padding=k // 2)
        dy = torch.randn_like(y)
        y.backward(dy)
        dw, db = ops.conv2d_wgrad(to(dev, _nhwc(x.detach())), to(dev, _nhwc(dy)), Cout, Cin, k, s_, want_bias=True)
        assert float((db.cpu() - dy.sum((0, 2, 3))).abs().max()) <= tol(dy.sum((0, 2, 3)).numpy())
        packed_t, rows, _ = ops.pack_conv_weight(to(dev, w.detach()), 1)
        dx = ops.conv2d_bwd_data(to(dev, _nhwc(dy)), packed_t, Cin, k, s_, (H, W))
        assert float((dw.cpu() - w.grad).abs().max()) <= tol(w.grad.numpy())
        assert float((nchw(dx.cpu()) - x.grad).abs().max()) <= tol(x.grad.numpy())
    # Cin 32 / 64 -> 64-row blocking, Cin 128 / 160 -> 128-row blocking of the per-phase launches; variant 0 = 64-row tiles
    for (B, H, W, Cin, Cout, force) in [(1, 5, 6, 32, 48, -1), (2, 13, 13, 64, 32, -1), (2, 13, 13, 128, 64, -1),
                                        (1, 9, 7, 160, 32, -1), (1, 9, 7, 160, 32, 0)]:
        x = torch.randn(B, Cin, H, W, requires_grad=True)
        wT = (torch.randn(Cin, Cout, 4, 4) * 0.1).requires_grad_()
        y = F.conv_transpose2d(x, wT, None, stride=2, padding=1)
        dy = torch.randn_like(y)
        y.backward(dy)
        _hip.call("dream_wgrad_set_variant", force)
        try:
            dw = ops.convT4x4_wgrad(to(dev, _nhwc(x.detach())), to(dev, _nhwc(dy)))
        finally:
            _hip.call("dream_wgrad_set_variant", -1)
        pk, rows = ops.pack_convT4x4_bwd_weight(to(dev, wT.detach()))
        dx = ops.conv4x4s2(to(dev, _nhwc(dy)), pk, rows)
        assert float((dw.cpu() - wT.grad).abs().max()) <= tol(wT.grad.numpy())
        assert float((nchw(dx.cpu()) - x.grad).abs().max()) <= tol(x.grad.numpy())
        other = torch.randn_like(x.detach())                 # a second gradient meeting at the same tensor
        dx2 = ops.conv4x4s2(to(dev, _nhwc(dy)), pk, rows, residual=to(dev, _nhwc(other)))
        assert float((nchw(dx2.cpu()) - (x.grad + other)).abs().max()) <= tol(x.grad.numpy())
    for xin in (torch.randn(2, 8, 9, 11), torch.round(torch.randn(1, 4, 7, 7) * 2)):     # second: ties
        xin.requires_grad_()
        y = F.max_pool2d(xin, 3, 2, 1)
        dy = torch.randn_like(y)
        y.backward(dy)
        dx = ops.maxpool3s2_bwd(to(dev, _nhwc(dy)), to(dev, _nhwc(xin.detach())))
        assert torch.equal(nchw(dx.cpu()), xin.grad)
        # the training pair: the forward pass stores the winner's window position, the backward pass gathers with it -- same bits
        yi, idx = ops.maxpool3s2_idx(to(dev, _nhwc(xin.detach())))
        assert torch.equal(nchw(yi.cpu()), y.detach()) and idx.dtype == torch.uint8 and int(idx.max()) <= 8
        dxi = ops.maxpool3s2_idx_bwd(to(dev, _nhwc(dy)), idx, tuple(_nhwc(xin.detach()).shape))
        assert torch.equal(dxi.cpu(), dx.cpu())
    a, b = torch.randn(1003), torch.randn(1003)
    c = to(dev, a.clone())
    ops.add_(c, to(dev, b))
    assert torch.equal(c.cpu(), a + b)
    cs = torch.randn(3, 5, 4, 64)
    assert float((ops.channel_sum(to(dev, cs)).cpu() - cs.sum((0, 1, 2))).abs().max()) < 1e-4


def check_vgg_train_grads(dev, arch="vgg_f", shape=(2, 32, 48)):
    """One training step of a VGG-type network against torch autograd on the CPU oracle (well-conditioned: no BN)."""
    k = cases.CNN_CASES[arch][0]
    b, h, w = shape
    wts = om.recipe_weights(om.build_model(arch, k).state_dict(), cases.TRAIN_FINAL_KEYS, cases.TRAIN_FINAL_SCALE)
    ref = om.build_model(arch, k)
    ref.load_state_dict(wts)
    ref.train()
    net = build_network(arch, dev, weights=wts, optimizer="sgd", lr=0.0, in_res=(w, h))
    net.enable_training()
    out_wh = net.net_output_resolution_from_input_resolution((w, h))
    x = torch.from_numpy(cases.image_batch(b, h, w, seed=3))
    t = torch.from_numpy(cases.target_batch(b, k, out_wh, in_wh=(w, h), seed=3))
    lref = F.mse_loss(ref(x)[0], t)
    lref.backward()
    loss = net.train([to(dev, x)], to(dev, t))
    assert abs(loss.item() - lref.item()) <= 1e-5 * abs(lref.item()) + 1e-9
    # One ReLU whose fp32 pre-activation lands on the other side of 0 than the oracle's (measured: 1 element in 98 304 at
    # this size) changes every upstream gradient by ~1e-3 of its maximum, so element-wise equality at 1e-5 is not a
    # property of a correct implementation.  Required instead: direction (cosine >= 0.9999) and a 2 % bound on the
    # worst element; a wiring / tap / transpose error fails both by orders of magnitude.
    for (name, p1), (_, p2) in zip(ref.named_parameters(), net.model.module.named_parameters()):
        g1, g2 = p1.grad.double(), p2.grad.cpu().double()
        scale = g1.abs().max().item() + 1e-12
        assert (g1 - g2).abs().max().item() <= 2e-2 * scale, name
        assert float((g1 * g2).sum() / (g1.norm() * g2.norm() + 1e-300)) >= 0.9999, name


def check_dataprep(dev):
    """On-device ToTensor+Normalize and create_belief_map are bit-identical to the reference semantics."""
    rs = np.random.RandomState(4)
    u8 = rs.randint(0, 256, (3, 37, 41, 3)).astype(np.uint8)
    mean, std = [0.5, 0.4, 0.6], [0.5, 0.25, 0.3]
    got = dream_amd.image_proc.normalize_images_u8(to(dev, torch.from_numpy(u8)), mean, std).cpu()
    ref = torch.from_numpy(u8).permute(0, 3, 1, 2).to(torch.float32).div(255)
    ref = (ref - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)     # torchvision Normalize
    assert torch.equal(got, ref)
    assert torch.equal(dream_amd.image_proc.normalize_images_u8(to(dev, torch.from_numpy(u8[:1])), [0.5] * 3, [0.5] * 3).cpu()[0],
                       torch.from_numpy(cases.image_batch(1, 37, 41, seed=4)[0]) * 0 + ((torch.from_numpy(u8[0]).permute(2, 0, 1).float() / 255 - 0.5) / 0.5))
    # every case of the reference-generated fixture (make_golden.py --only-belief-maps), per-frame drop-in and batched
    gold = np.load(os.path.join(GOLD, "belief_map_golden.npz"))
    for name, (res, pts, sigma) in cases.belief_map_cases().items():
        ref64 = gold[name]
        one = dream_amd.image_proc.create_belief_map(res, [tuple(p) for p in pts], sigma=sigma)
        assert one.dtype == np.float64 and one.shape == ref64.shape, name
        # the reference's only consumer keeps torch.tensor(maps).float() (datasets.py:165-171)
        assert np.array_equal(one.astype(np.float32), ref64.astype(np.float32)), name
        got = dream_amd.image_proc.create_belief_map_batch(res, to(dev, torch.from_numpy(np.stack([pts, pts[::-1]]))), sigma).cpu()
        assert torch.equal(got[0], torch.tensor(ref64).float()) and torch.equal(got[1], torch.tensor(ref64[::-1].copy()).float()), name
    kps = np.array([[[65.0, 20.0], [100.0, 80.0], [4.0, 4.0], [3.9, 10.0], [74.9, 55.2], [75.0, 30.0], [-0.5, 20.0]],
                    [[10.2, 10.7], [40.0, 54.0], [40.0, 55.0], [12.0, 4.0], [12.0, 3.99], [79.0, 59.0], [30.5, 30.5]]], np.float32)
    got = dream_amd.image_proc.create_belief_map_batch((80, 60), to(dev, torch.from_numpy(kps))).cpu()
    for b in range(2):
        ref = torch.tensor(op.create_belief_map((80, 60), kps[b])).float()
        assert torch.equal(got[b], ref), b
    assert float(got[0, 0].max()) == 1.0 and float(got[0, 1].abs().max()) == 0.0 and float(got[0, 6].abs().max()) == 0.0
    one = dream_amd.image_proc.create_belief_map((80, 60), [tuple(p) for p in kps[1]])       # the reference's per-frame call
    assert one.dtype == np.float64 and one.shape == (7, 60, 80)
    assert np.array_equal(one.astype(np.float32), op.create_belief_map((80, 60), kps[1]).astype(np.float32))


def check_conv_transpose4x4_f16x3(dev, B, H, W, Cin, Cout, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    x[0, 0, 0, 0] = 30.0
    wT = torch.randn(Cin, Cout, 4, 4, generator=g) * (2.0 / (4 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x.double(), wT.double(), bias.double(), stride=2, padding=1).relu()
    p16 = ops.pack_convT4x4_weight_f16x3(to(dev, wT))
    y, amax = ops.conv_transpose4x4s2_f16x3(to(dev, _nhwc(x)), ops.absmax(to(dev, x)), p16, p16[3], None, to(dev, bias), ops.CONV_RELU)
    got = y.cpu().permute(0, 3, 1, 2)
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) / scale <= 5e-6
    am = float(np.frombuffer(amax.cpu().numpy().tobytes(), dtype=np.float32)[0])
    assert abs(am - float(got.abs().max())) <= 1e-6 * scale


def check_variant(dev, name, precision="fp32", train=True):
    """Hourglass constructor branches outside the shipped YAMLs (oracle.models.VARIANTS) against the reference's golden
    outputs: inference (last stage maps + keypoints), every head / stage of model(x), and two Adam steps."""
    shapes, has_train = cases.VARIANT_CASES[name]
    g = np.load(os.path.join(GOLD, "variant_%s.npz" % name))
    net = build_network(name, dev)
    assert list(net.model.state_dict().keys()) == ["module." + k for k in om.build_model(name, 7).state_dict().keys()]
    net.enable_evaluation()
    net.model.module.precision = precision
    for (b, h, w) in shapes:
        tag = "%dx%dx%d" % (b, h, w)
        x = to(dev, torch.from_numpy(cases.image_batch(b, h, w, seed=b * 1000 + h)))
        with torch.no_grad():
            res = net.inference(x)
            heads = net.model(x)
        for i, t in enumerate(heads):
            ref = g[tag + "/head%d" % i]
            lim = tol(ref) if ref.ndim == 4 else 2e-2          # soft-argmax coordinates (px) amplify 1e-5 map noise
            assert np.abs(t.cpu().numpy() - ref).max() <= lim, (name, tag, i, np.abs(t.cpu().numpy() - ref).max())
        assert np.abs(res[0].cpu().numpy() - g[tag + "/maps"]).max() <= tol(g[tag + "/maps"])
        ref_k, got_k = g[tag + "/keypoints"], res[1].cpu().numpy()
        if ref_k.dtype == got_k.dtype and name != "vgg_q_softmax":
            same = (got_k == np.float32(-999.999)) == (ref_k == np.float32(-999.999))
            assert same.mean() >= 0.85
            both = (got_k != np.float32(-999.999)) & (ref_k != np.float32(-999.999))
            assert np.abs(got_k - ref_k)[both].max(initial=0.0) < 0.5
        else:
            assert np.abs(got_k - ref_k).max() <= 2e-2
    if not (train and has_train and precision == "fp32"):
        return
    b, h, w = cases.VARIANT_TRAIN_SHAPE
    wts = om.recipe_weights(om.build_model(name, 7).state_dict(), cases.TRAIN_FINAL_KEYS, cases.TRAIN_FINAL_SCALE)
    net = build_network(name, dev, weights=wts, optimizer="adam", lr=cases.TRAIN_LR["adam"], in_res=(w, h))
    net.enable_training()
    ow, oh = net.trained_net_output_resolution()
    x = to(dev, torch.from_numpy(cases.image_batch(b, h, w, seed=7)))
    t = to(dev, torch.from_numpy(cases.target_batch(b, 7, (ow, oh), in_wh=(w, h), seed=7)))
    losses = []
    for step in range(2):
        losses.append(net.train([x], t).item())
        if step == 0:
            for key, p in net.model.named_parameters():
                ref = float(g["train/gradnorm/" + key])
                assert abs(float(p.grad.double().norm()) - ref) <= 2e-3 * max(ref, 1e-9), (key, float(p.grad.double().norm()), ref)
    assert np.allclose(losses, g["train/losses"], rtol=2e-4), (losses, g["train/losses"])
    # Adam moves every element by ~lr per step whatever the gradient's size, so an element whose gradient is rounding
    # noise around zero may step the other way: bound every element by that worst case, and require the bulk to agree.
    lr, close, total = cases.TRAIN_LR["adam"], 0, 0
    for key, p in net.model.named_parameters():
        s_ = p.detach().flatten()[:: max(1, p.numel() // 64)][:64].cpu().numpy()
        ref = g["train/param_sample/" + key]
        assert np.abs(s_ - ref).max() <= 2 * 2 * lr * 1.05 + 1e-3 * np.abs(ref).max(), key
        close += int(np.isclose(s_, ref, rtol=1e-3, atol=3e-6).sum())
        total += ref.size
    assert close >= 0.97 * total, (close, total)


def check_keypoint_conversions(dev):
    """Host drop-ins and the batched device kernel against the reference's outputs (keypoint_conversion.npz), bit for bit."""
    g = np.load(os.path.join(GOLD, "keypoint_conversion.npz"))
    ip = dream_amd.image_proc
    for name, (kps, out_res, in_res, raw_res) in cases.keypoint_conversion_cases().items():
        netin = ip.convert_keypoints_to_netin_from_netout(kps.astype(float), out_res, in_res)
        assert np.array_equal(netin, g[name + "/netin"]), name
        for prep in ("none", "resize", "shrink", "shrink-and-crop"):
            want = g[name + "/raw/" + prep]
            assert np.array_equal(ip.convert_keypoints_to_raw_from_netin(netin, in_res, raw_res, prep), want), (name, prep)
            d_in, d_raw = ip.convert_keypoints_batch(to(dev, torch.from_numpy(kps)), out_res, in_res, raw_res, prep)
            assert np.array_equal(d_in.cpu().numpy(), g[name + "/netin"]) and np.array_equal(d_raw.cpu().numpy(), want), (name, prep)


def check_conv_transpose3x3(dev, B, H, W, Cin, Cout, relu=True, seed=0):
    """Sub-pixel ConvTranspose2d(3,2,1,output_padding 1) against torch, and against the zero-stuffed form bit for bit
    in exact arithmetic terms (same products, summed per output in tap order within a chunk)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    wT = torch.randn(Cin, Cout, 3, 3, generator=g) * (2.0 / (2.25 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x, wT, bias, stride=2, padding=1, output_padding=1)
    if relu:
        ref = ref.relu()
    packed, rows, _, _ = ops.pack_weight(to(dev, wT), 1)
    y = ops.conv_transpose3x3s2(to(dev, _nhwc(x)), packed, to(dev, bias), rows, relu=relu)
    assert tuple(y.shape) == (B, 2 * H, 2 * W, Cout)
    assert float((y.cpu().permute(0, 3, 1, 2) - ref).abs().max()) <= tol(ref.numpy())
    z = ops.conv3x3(to(dev, _nhwc(x)), packed, to(dev, bias), rows, (ops.CONV_RELU if relu else 0) | ops.CONV_ZEROSTUFF2X)
    assert float((y - z).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


def check_upsample_conv_as_convT(dev, B, H, W, Cin, Cout, relu=True, seed=0):
    """nn.Upsample(2) + Conv2d(k3,p1) (+ReLU) through the equivalent 4x4 transposed conv, against torch and against the
    DREAM_CONV_UPSAMPLE2X form of the same kernel."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.interpolate(x, scale_factor=2), w, bias, padding=1)
    if relu:
        ref = ref.relu()
    pk4, cout = ops.pack_convT4x4_weight(ops.upsample_conv_weight(to(dev, w)))
    y = ops.conv_transpose4x4s2(to(dev, _nhwc(x)), pk4, cout, None, to(dev, bias), ops.CONV_RELU if relu else 0)
    assert float((y.cpu().permute(0, 3, 1, 2) - ref).abs().max()) <= tol(ref.numpy())
    packed, rows, _, _ = ops.pack_weight(to(dev, w), 0)
    z = ops.conv3x3(to(dev, _nhwc(x)), packed, to(dev, bias), rows, (ops.CONV_RELU if relu else 0) | ops.CONV_UPSAMPLE2X)
    assert float((y - z).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))       # two fp32 summation orders


def check_wgrad(dev, B, H, W, Cin, Cout, k=3, stride=1, seed=0):
    """Weight + bias gradient of a k x k / stride conv against torch autograd."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).requires_grad_()
    y = F.conv2d(x, w, None, stride=stride, padding=k // 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw, db = ops.conv2d_wgrad(to(dev, _nhwc(x)), to(dev, _nhwc(dy)), Cout, Cin, k, stride, want_bias=True)
    assert float((dw.cpu() - w.grad).abs().max()) <= tol(w.grad.numpy()), (B, H, W, Cin, Cout, k, stride)
    assert float((db.cpu() - dy.sum((0, 2, 3))).abs().max()) <= tol(dy.sum((0, 2, 3)).numpy())


def check_conv3x3_bn_fused(dev, cases=None):
    """Train-mode BatchNorm folded into the stride-1 3x3 conv's launches on the Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip
    WINO_STAT): conv -> statistics of the FOLLOWING BatchNorm in the epilogue; data gradient -> ReLU mask of the PRECEDING BatchNorm
    recomputed from (z, ab) + its two backward reductions.  Against the unfused launches (same conv bits, same mask) and against
    torch's conv2d / BatchNorm2d / autograd on CPU; the ticket words are left zero."""
    def nchw(t):
        return t.permute(0, 3, 1, 2)

    def make_bn(C):
        bn = torch.nn.BatchNorm2d(C)
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_(0.0, 0.5)
        twin = torch.nn.BatchNorm2d(C)
        twin.load_state_dict(bn.state_dict())
        return bn, (twin.to(dev) if dev != "cpu" else twin)

    torch.manual_seed(23)
    # (B, H, W, Cin, Cout): one and two column blocks, the 4- and the 8-wave kernel, odd extents (half tiles), more tile blocks than
    # workgroups (several blocks per producer row) and fewer (producer rows of zeros)
    cases = cases or [(2, 9, 11, 64, 64), (1, 13, 13, 64, 128), (3, 6, 5, 128, 256), (2, 25, 25, 64, 128), (1, 4, 4, 64, 64)]
    for (B, H, W, Cin, Cout) in cases:
        bn_p, bn_p2 = make_bn(Cin)               # the BatchNorm in front of the conv (bn1 of a Bottleneck)
        bn_n, bn_n2 = make_bn(Cout)              # the one behind it (bn2)
        z1 = (torch.randn(B, Cin, H, W) + 0.2).requires_grad_()
        w = (torch.randn(Cout, Cin, 3, 3) * (2.0 / (9 * Cin)) ** 0.5).requires_grad_()
        y1 = bn_p(z1).relu()
        z2_ref = F.conv2d(y1, w, None, padding=1)
        y2_ref = bn_n(z2_ref).relu()
        dy2 = torch.randn_like(y2_ref)
        y2_ref.backward(dy2)
        ctr = to(dev, torch.zeros(4096, dtype=torch.int32))
        z1_d = to(dev, _nhwc(z1.detach()))
        ab1, mean1, invstd1 = ops.bn_stats(z1_d, bn_p2, ctr)
        y1_d = ops.bn_apply_ab(z1_d, ab1, None, True)
        u, rows = ops.pack_weight_winograd_tile(to(dev, w.detach()), 0, 2)
        # ---- forward: conv + statistics in one launch
        z2, ab2, mean2, invstd2 = ops.conv3x3_winograd_bn(y1_d, u, rows, bn_n2, ctr)
        z2_plain = ops.conv3x3_winograd(y1_d, u, rows)
        assert torch.equal(z2, z2_plain), (B, H, W, Cin, Cout)
        assert float((nchw(z2.cpu()) - z2_ref.detach()).abs().max()) < 2e-5 * float(z2_ref.abs().max())
        assert float((mean2.cpu() - z2_ref.detach().mean((0, 2, 3))).abs().max()) < 1e-5
        var_ref = z2_ref.detach().var((0, 2, 3), unbiased=False)
        assert float((invstd2.cpu() - (var_ref + bn_n.eps).rsqrt()).abs().max()) < 1e-4 * float((var_ref + bn_n.eps).rsqrt().max())
        assert float((ab2[0].cpu() - bn_n.weight.detach() * invstd2.cpu()).abs().max()) < 1e-6
        assert float((ab2[1].cpu() - (bn_n.bias.detach() - mean2.cpu() * ab2[0].cpu())).abs().max()) < 1e-6
        assert float((bn_n.running_mean - bn_n2.running_mean.cpu()).abs().max()) < 1e-6
        assert float((bn_n.running_var - bn_n2.running_var.cpu()).abs().max()) < 1e-5
        assert int(bn_n2.num_batches_tracked.item()) == 1
        # the stand-alone statistics of the same tensor agree to round-off (fp64 rows of different shapes)
        twin = torch.nn.BatchNorm2d(Cout)
        twin.load_state_dict(bn_n.state_dict())
        twin.running_mean.zero_(); twin.running_var.fill_(1.0); twin.num_batches_tracked.zero_()
        twin = twin.to(dev) if dev != "cpu" else twin
        ab_s, mean_s, invstd_s = ops.bn_stats(z2, twin, ctr)
        assert float((mean_s - mean2).abs().max()) < 1e-6 and float((invstd_s / invstd2 - 1).abs().max()) < 1e-5
        # ---- backward: data gradient of the conv + mask and reductions of bn_p in one launch
        y2 = ops.bn_apply_ab(z2, ab2, None, True)
        dy2_d = to(dev, _nhwc(dy2))
        dgam2, dbet2 = ops.bn_bwd_stats(z2, dy2_d, mean2, invstd2, ctr, y_act=y2)
        dz2, _ = ops.bn_bwd_apply(z2, dy2_d, bn_n2.weight, mean2, invstd2, dgam2, dbet2, y_act=y2)
        u_t, rows_t = ops.pack_weight_winograd_tile(to(dev, w.detach()), 1, 2)
        g, dgam1, dbet1 = ops.conv3x3_winograd_bwd_bnmask(dz2, u_t, Cin, z1_d, ab1, mean1, invstd1, ctr)
        g_raw = ops.conv3x3_winograd(dz2, u_t, rows_t)
        dgam_u, dbet_u = ops.bn_bwd_stats(z1_d, g_raw, mean1, invstd1, ctr, ab=ab1)
        dz1_u, g_u = ops.bn_bwd_apply(z1_d, g_raw, bn_p2.weight, mean1, invstd1, dgam_u, dbet_u, ab=ab1, want_g=True)
        assert torch.equal(g, g_u), (B, H, W, Cin, Cout)                      # same conv bits, same recomputed mask
        scale = max(float(dgam_u.abs().max()), float(dbet_u.abs().max()), 1e-3)
        assert float((dgam1 - dgam_u).abs().max()) < 1e-5 * scale and float((dbet1 - dbet_u).abs().max()) < 1e-5 * scale
        dz1, _ = ops.bn_bwd_apply(z1_d, g, bn_p2.weight, mean1, invstd1, dgam1, dbet1)     # g is masked already: no mask source
        # against autograd through torch's own conv / BatchNorm: two Winograd convs and two BatchNorm backward passes of round-off
        # (a wrong mask or sum is an O(1) error).  The mask sits on a*z + b > 0 evaluated in two arithmetics: among millions of
        # elements one may land within round-off of zero and flip (measured on the GPU: one element of 5.1 M, an error of 1.7e-2 of
        # the maximum there and an O(1) term in dbeta) -- so on large tensors the bound (1e-3) is on all but a 1e-4 fraction of the
        # elements, and the sums are held to the unfused launches above (same mask) instead of to torch.
        rel = (nchw(dz1.cpu()) - z1.grad).abs() / max(1.0, float(z1.grad.abs().max()))
        if B * H * W <= 20000:
            assert float(rel.max()) < 2e-4, ((B, H, W, Cin, Cout), float(rel.max()))
            assert float((dgam1.cpu() - bn_p.weight.grad).abs().max()) < 1e-4 * max(1.0, float(bn_p.weight.grad.abs().max()))
            assert float((dbet1.cpu() - bn_p.bias.grad).abs().max()) < 1e-4 * max(1.0, float(bn_p.bias.grad.abs().max()))
        else:
            assert float((rel > 1e-3).float().mean()) < 1e-4, ((B, H, W, Cin, Cout), float((rel > 1e-3).float().mean()))
        assert int(ctr.cpu().abs().sum()) == 0       # every launch leaves its ticket words zero
    return True


def check_bn_fused_ops(dev, ksplits=(0, 1, 2, 4)):
    """Round 4: train-mode BatchNorm without its separate passes (csrc/bn.hip "round 4", csrc/gemm1x1.hip PRE / EPI) against
    torch's BatchNorm2d / conv2d / autograd on CPU: statistics finished inside the launch (stand-alone and in the GEMM epilogue),
    BN + ReLU applied by the consumer's loader, the backward reductions in the data-gradient epilogue, the ticket words left zero."""
    def nchw(t):
        return t.permute(0, 3, 1, 2)

    def make_bn(C):
        bn = torch.nn.BatchNorm2d(C)
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_(0.0, 0.5)
        twin = torch.nn.BatchNorm2d(C)
        twin.load_state_dict(bn.state_dict())
        return bn, (twin.to(dev) if dev != "cpu" else twin)

    torch.manual_seed(11)
    # ---- stand-alone: statistics in one launch, apply from (a, b), backward with the three mask sources ------------------------
    for (B, C, H, W, relu, res) in [(2, 64, 5, 7, True, True), (3, 256, 4, 4, True, False), (2, 2048, 3, 3, False, False),
                                    (1, 48, 6, 5, True, False), (3, 128, 33, 41, True, False)]:
        bn, bn2 = make_bn(C)
        ctr = to(dev, torch.zeros(4096, dtype=torch.int32))
        x = (torch.randn(B, C, H, W) * 1.5 + 0.3).requires_grad_()
        r = torch.randn(B, C, H, W) if res else None
        y_ref = bn(x)
        if res:
            y_ref = y_ref + r
        if relu:
            y_ref = y_ref.relu()
        dy = torch.randn_like(y_ref)
        y_ref.backward(dy)
        z = to(dev, _nhwc(x.detach()))
        versions = [t._version for t in (bn2.running_mean, bn2.running_var, bn2.num_batches_tracked)]
        ab, mean, invstd = ops.bn_stats(z, bn2, ctr)
        assert all(t._version > v for t, v in zip((bn2.running_mean, bn2.running_var, bn2.num_batches_tracked), versions))
        y = ops.bn_apply_ab(z, ab, to(dev, _nhwc(r)) if res else None, relu)
        assert float((nchw(y.cpu()) - y_ref).abs().max()) < 1e-5
        assert float((bn.running_mean - bn2.running_mean.cpu()).abs().max()) < 1e-6
        assert float((bn.running_var - bn2.running_var.cpu()).abs().max()) < 1e-5
        assert int(bn2.num_batches_tracked.item()) == 1
        # the published affine map is BatchNorm: a = gamma * invstd, b = beta - mean * a
        assert float((ab[0].cpu() - bn.weight.detach() * invstd.cpu()).abs().max()) < 1e-6
        sources = [dict(y_act=y)] if relu else [dict()]
        if relu and not res:
            sources.append(dict(ab=ab))             # mask recomputed from the BatchNorm input
        for src in sources:
            dgam, dbet = ops.bn_bwd_stats(z, to(dev, _nhwc(dy)), mean, invstd, ctr, **src)
            dx, g = ops.bn_bwd_apply(z, to(dev, _nhwc(dy)), bn2.weight, mean, invstd, dgam, dbet, want_g=True, **src)
            assert float((nchw(dx.cpu()) - x.grad).abs().max()) < 1e-5, (C, src.keys())
            assert float((dgam.cpu() - bn.weight.grad).abs().max()) < 1e-4 and float((dbet.cpu() - bn.bias.grad).abs().max()) < 1e-4
            gm = dy * (y_ref > 0) if relu else dy
            if "ab" in src:                          # a value within round-off of zero may flip: none here, and never silently many
                assert float((nchw(g.cpu()) != gm).float().mean()) < 1e-4
            else:
                assert float((nchw(g.cpu()) - gm).abs().max()) == 0.0
        assert int(ctr.cpu().abs().sum()) == 0       # every launch leaves its ticket words zero
    # ---- the 1x1 conv with the BatchNorm on either side folded in --------------------------------------------------------------------
    for (B, H, W, Cin, Cout, with_pre, with_bias) in [(2, 9, 11, 64, 96, True, False), (1, 13, 13, 128, 256, True, True),
                                                      (3, 5, 5, 256, 64, False, False), (2, 7, 6, 32, 160, True, False),
                                                      (2, 9, 11, 64, 32, True, True),           # 32 output channels: the data gradient contracts over K = 32
                                                                                                 # (the head conv's zero-padded keypoint channels, round 6)
                                                      (3, 33, 41, 64, 64, True, False),         # 64 row blocks: both levels of the ticket tree
                                                      (4, 65, 66, 32, 64, True, False)]:        # 269 row blocks: groups of 17 rows (> one round trip)
        bn_p, bn_p2 = make_bn(Cin)
        bn_n, bn_n2 = make_bn(Cout)
        zp = (torch.randn(B, Cin, H, W) + 0.2).requires_grad_()
        w = (torch.randn(Cout, Cin, 1, 1) * 0.1).requires_grad_()
        bias = torch.randn(Cout) * 0.1 if with_bias else None
        xin = bn_p(zp).relu() if with_pre else zp
        z_ref = F.conv2d(xin, w, bias)
        y_ref = bn_n(z_ref).relu()
        dyo = torch.randn_like(y_ref)
        y_ref.backward(dyo)
        ctr = to(dev, torch.zeros(4096, dtype=torch.int32))
        zp_d = to(dev, _nhwc(zp.detach()))
        pre_ab = pre_mean = pre_invstd = None
        if with_pre:
            pre_ab, pre_mean, pre_invstd = ops.bn_stats(zp_d, bn_p2, ctr)
        packed, rows = ops.pack_conv1x1_weight(to(dev, w.detach()), 0)
        packed_t, rows_t = ops.pack_conv1x1_weight(to(dev, w.detach()), 1)
        for ks, rows_forced in [(k_, r_) for k_ in ksplits for r_ in ((0, 64, 32) if k_ else (0,))]:
            if ks and Cin % (32 * ks):
                continue
            bn_k, bn_k2 = make_bn(Cout)
            bn_k2.load_state_dict({k_: v.to(bn_k2.weight.device) for k_, v in bn_n.state_dict().items()})
            bn_k2.running_mean.zero_(); bn_k2.running_var.fill_(1.0); bn_k2.num_batches_tracked.zero_()
            _hip.call("dream_conv1x1_set_ksplit", ks)
            _hip.call("dream_conv1x1_set_rows", rows_forced)
            try:
                z, ab, mean, invstd = ops.conv1x1_bn(zp_d, packed, rows, bn_k2, ctr, pre_ab=pre_ab, shift=to(dev, bias) if with_bias else None)
            finally:
                _hip.call("dream_conv1x1_set_ksplit", 0)
                _hip.call("dream_conv1x1_set_rows", 0)
            assert float((nchw(z.cpu()) - z_ref.detach()).abs().max()) <= tol(z_ref.detach().numpy()), (Cin, Cout, ks)
            y = ops.bn_apply_ab(z, ab, None, True)
            assert float((nchw(y.cpu()) - y_ref.detach()).abs().max()) < 2e-5, (Cin, Cout, ks)
            assert float((bn_n.running_mean - bn_k2.running_mean.cpu()).abs().max()) < 1e-5
            assert float((bn_n.running_var - bn_k2.running_var.cpu()).abs().max()) < 1e-5
            assert int(bn_k2.num_batches_tracked.item()) == 1 and int(ctr.cpu().abs().sum()) == 0
        # backward of the following BatchNorm from the stored activation, then this conv's data gradient with the PREVIOUS BatchNorm's
        # ReLU mask and reductions in its epilogue, and the weight gradient through the loader-side BatchNorm
        dgam, dbet = ops.bn_bwd_stats(z, to(dev, _nhwc(dyo)), mean, invstd, ctr, y_act=y)
        dz, _ = ops.bn_bwd_apply(z, to(dev, _nhwc(dyo)), bn_n2.weight, mean, invstd, dgam, dbet, y_act=y)
        assert float((dgam.cpu() - bn_n.weight.grad).abs().max()) < 2e-4 * max(1.0, float(bn_n.weight.grad.abs().max()))
        dw = ops.conv1x1_wgrad(zp_d, dz, Cout, Cin, pre_ab=pre_ab) if (with_pre and Cin % 64 == 0) else None
        if dw is not None:
            assert float((dw.cpu() - w.grad).abs().max()) <= 3 * tol(w.grad.numpy()), (Cin, Cout)
        if with_pre:
            for ks, rows_forced in [(k_, r_) for k_ in ksplits for r_ in ((0, 64, 32) if k_ else (0,))]:
                if ks and Cout % (32 * ks):
                    continue
                _hip.call("dream_conv1x1_set_ksplit", ks)
                _hip.call("dream_conv1x1_set_rows", rows_forced)
                try:
                    gm, dg_p, db_p = ops.conv1x1_bwd_bnmask(dz, packed_t, Cin, zp_d, pre_ab, pre_mean, pre_invstd, ctr)
                    if ks == 0:      # the same with the mask read from the stored activation (+ a second gradient meeting there)
                        extra = to(dev, _nhwc(torch.zeros_like(zp.detach())))
                        gm2, dg2, db2 = ops.conv1x1_bwd_bnmask(dz, packed_t, Cin, zp_d, None, pre_mean, pre_invstd, ctr, y_act=ops.bn_apply_ab(zp_d, pre_ab, None, True), residual=extra)
                        assert torch.equal(gm2, gm) and torch.equal(dg2, dg_p) and torch.equal(db2, db_p)
                finally:
                    _hip.call("dream_conv1x1_set_ksplit", 0)
                    _hip.call("dream_conv1x1_set_rows", 0)
                assert float((dg_p.cpu() - bn_p.weight.grad).abs().max()) < 3e-4 * max(1.0, float(bn_p.weight.grad.abs().max())), (Cin, ks)
                assert float((db_p.cpu() - bn_p.bias.grad).abs().max()) < 3e-4 * max(1.0, float(bn_p.bias.grad.abs().max()))
                dzp, _ = ops.bn_bwd_apply(zp_d, gm, bn_p2.weight, pre_mean, pre_invstd, dg_p, db_p)
                assert float((nchw(dzp.cpu()) - zp.grad).abs().max()) <= 3 * tol(zp.grad.numpy()), (Cin, Cout, ks)
                assert int(ctr.cpu().abs().sum()) == 0


def check_peaks_fused_equals_three_kernels(dev, sizes=((1, 7, 5, 9), (2, 3, 37, 45), (1, 2, 64, 64), (2, 7, 100, 100), (1, 2, 130, 71))):
    """The keypoint rule through the fused row pass + scan (round 6, csrc/peaks.hip: gauss_row_peaks_kernel + peaks_finish_kernel) against the
    three-kernel form: keypoints and peak counts bit for bit, on smooth random maps with several peaks (ties and the next-best rule included)."""
    rs = np.random.RandomState(11)
    most = 0
    for b, k, h, w in sizes:
        yy, xx = np.mgrid[0:h, 0:w]
        maps = np.zeros((b, k, h, w), np.float32)
        for i in range(b):
            for j in range(k):
                for _ in range(int(rs.randint(0, 4))):
                    cy, cx, s, a = rs.uniform(0, h), rs.uniform(0, w), rs.uniform(1.0, 4.0), rs.uniform(0.2, 1.0)
                    maps[i, j] += (a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))).astype(np.float32)
        maps[0, 0] = maps[0, 0].T.copy().T if h == w else maps[0, 0]
        maps += (rs.rand(b, k, h, w) < 0.002).astype(np.float32) * 0.5            # isolated spikes: more peaks, equal scores
        m = to(dev, torch.from_numpy(maps))
        outs = []
        for fused in (1, 0):
            _hip.call("dream_peaks_set_fused", fused)
            try:
                for use_scores, nb in ((True, 0.25), (True, 0.0), (False, 0.25)):
                    kp, cnt = ops.keypoints_from_belief_maps(m, 0.0 if h % 2 else 0.5, use_scores, nb)
                    outs.append((fused, kp.cpu().clone(), cnt.cpu().clone()))
            finally:
                _hip.call("dream_peaks_set_fused", -1)
        half = len(outs) // 2
        for (f1, kp1, c1), (f0, kp0, c0) in zip(outs[:half], outs[half:]):
            assert torch.equal(c1, c0), (b, k, h, w, c1.tolist(), c0.tolist())
            assert kp1.numpy().tobytes() == kp0.numpy().tobytes(), (b, k, h, w)
        most = max(most, int(outs[0][2].max()))
    assert most >= 2                                                               # the case families really have maps with several peaks
