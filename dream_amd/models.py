"""MI355X-native belief-map networks behind the reference's model interface.

Mirrors /root/reference/dream/models.py:
  * ``DreamHourglass``  (models.py:557-827)  -- VGG19 encoder + upsample (Q) or ConvTranspose (F)
    decoder + 3-conv head; ``state_dict()`` keys/shapes are the reference's, so released ``.pth``
    files and ``dream_network.model.load_state_dict(torch.load(path))`` (dream/analysis.py:148) work.
  * ``ResnetSimple``    (models.py:17-155)   -- torchvision ResNet-101 trunk + ConvTranspose/BN decoder, evaluation
    (BatchNorm folded) and training (batch statistics) on the same kernels; identical ``state_dict()``.
  * ``DreamHourglassMultiStage`` (models.py:350-553) -- 1..6 hourglasses, each fed the image and the previous maps.
  * ``DreamDataParallel`` (dream_amd/data_parallel.py) stands in for ``torch.nn.DataParallel`` (dream/network.py:244-256):
    the ``module.`` key prefix, and persistent single-process replicas on the GPUs listed in ``gpu_ids``; under torchrun it is
    a pass-through and the gradient exchange is the bucketed RCCL all-reduce in ``_HourglassFunction.backward`` below.

``nn.Conv2d`` objects are used purely as parameter containers (OIHW, as the reference stores them);
their ATen forward is never called.  ``forward`` executes a static list of C-ABI calls on NHWC
activations: first conv (NCHW image -> NHWC, VALU), 3x3 convs on the fp32 matrix cores (Winograd F(2x2,3x3) where it
applies, direct implicit GEMM elsewhere) with fused bias/ReLU/upsample/max-pool,
skip-connection adds, and an NCHW store in the last head conv; ``backward`` walks the same list in reverse.  If the HIP library or a GPU is missing the
call raises -- there is no CPU path in this package.
"""
import os
import threading

import torch
import torch.nn as nn

from . import ops, pretrained as _pretrained
from .ops import CONV_RELU, CONV_UPSAMPLE2X, CONV_OUT_NCHW, CONV_ZEROSTUFF2X, CONV_POOL2
from .spatial_softmax import SoftArgmaxPavlo

# (container name, [(child index, cin, cout)]) -- child indices are torchvision's vgg19.features
# indices the reference re-uses (models.py:598-615), which is what fixes the state_dict keys.
_ENCODER = [
    ("layer_0_1_down", [(0, 3, 64), (2, 64, 64)]),
    ("layer_0_2_down", [(5, 64, 128), (7, 128, 128)]),
    ("layer_0_3_down", [(10, 128, 256), (12, 256, 256), (14, 256, 256), (16, 256, 256)]),
    ("layer_0_4_down", [(19, 256, 512), (21, 512, 512), (23, 512, 512), (25, 512, 512)]),
    ("layer_0_5_down", [(28, 512, 512), (30, 512, 512), (32, 512, 512), (34, 512, 512)]),
]


class _Params(nn.Sequential):
    """Sequential used only as a named parameter container (children are never called)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("dream_amd: parameter container, not callable; use the parent module")


class _PackedCache:
    """Packed copies of one conv weight for the MFMA kernels, refreshed when the parameter changes
    (optimizer step / load_state_dict bump ``_version``)."""

    def __init__(self):
        self._store = {}

    def get(self, weight, mode, f16x3=False):
        """mode 0 / 1: forward / transposed tap-major packing; "ups": the conv that follows a nearest x2 upsample, as the
        equivalent 4x4 transposed conv (ops.upsample_conv_weight) in the sub-pixel phase packing."""
        key = (id(weight), mode, f16x3)
        tag = (weight._version, weight.data_ptr(), weight.device)
        hit = self._store.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                if mode in ("wino0", "wino1"):                    # Winograd F(2x2,3x3) transformed weights (fwd / data gradient)
                    packed = ops.pack_weight_winograd(weight.detach(), int(mode[-1]))
                elif mode in ("wino4_0", "wino4_1"):              # Winograd F(4x4,3x3)
                    packed = ops.pack_weight_winograd4(weight.detach(), int(mode[-1]))
                elif mode == "ups":
                    wt4 = ops.upsample_conv_weight(weight.detach())
                    packed = ops.pack_convT4x4_weight_f16x3(wt4) if f16x3 else ops.pack_convT4x4_weight(wt4)
                elif mode in ("ups_wino0", "ups_wino1"):          # the same transposed conv by minimal filtering (fwd / data gradient)
                    packed = ops.pack_convT4x4_winograd_weight(ops.upsample_conv_weight(weight.detach()), int(mode[-1]))
                elif mode in ("ups_wino4_0", "ups_wino4_1"):      # ... on the F(4x4,3x3) kernel (25-position phase patterns)
                    packed = ops.pack_convT4x4_winograd4_weight(ops.upsample_conv_weight(weight.detach()), int(mode[-1]))
                else:
                    packed = ops.pack_conv_weight_f16x3(weight.detach(), mode) if f16x3 else ops.pack_weight(weight.detach(), mode)
                hit = (tag, packed)
            self._store[key] = hit
        return hit[1]


class DreamHourglass(nn.Module):
    # input pixels per step up to which weight gradients run on a second stream: +1.4 % at 16 frames of 400x400, -0.2 % at
    # 32 (the VGG layers are large enough to fill the chip on their own; see ResnetSimple for the case where it pays)
    OVERLAP_MAX_PIXELS = int(os.environ.get("DREAM_VGG_OVERLAP_MAX_FRAMES", "16")) * 400 * 400

    def __init__(self, n_keypoints, n_image_input_channels=3, internalize_spatial_softmax=True,
                 learned_beta=True, initial_beta=1.0, skip_connections=False, deconv_decoder=False,
                 full_output=False):
        super().__init__()
        self.n_keypoints = n_keypoints
        self.n_image_input_channels = n_image_input_channels
        self.internalize_spatial_softmax = internalize_spatial_softmax
        self.skip_connections = skip_connections
        self.deconv_decoder = deconv_decoder
        self.full_output = full_output
        if internalize_spatial_softmax:
            self.n_output_heads = 2
            self.learned_beta = learned_beta
            self.initial_beta = initial_beta
        else:
            self.n_output_heads = 1
            self.learned_beta = False
        # plan entries: (kind, container name, child name, flags).  kinds: "first" (VALU conv on the NCHW image, <= 4
        # channels), "wide" (first conv of a multi-stage hourglass: image + previous maps, NHWC zero-padded to the MFMA
        # kernel's channel granularity), "conv", "deconv", "pool", and "add" (skip connection; flags = plan index of the
        # entry whose output is added, models.py:774-799).
        def conv(ci, co):
            return nn.Conv2d(ci, co, kernel_size=3, stride=1, padding=1)

        def deconv(ci, co):
            return nn.ConvTranspose2d(ci, co, kernel_size=(3, 3), stride=(2, 2), padding=1, output_padding=1)

        def container(items):
            seq = _Params()
            for name, mod in items:
                seq.add_module(str(name), mod)
            return seq

        plan = []
        x1 = None                                            # plan index of x_0_1 (last conv of the first block)
        pooled = {}                                          # block k -> plan index of x_0_k_d
        for bi, (cname, convs) in enumerate(_ENCODER):
            items = []
            for (idx, ci, co) in convs:
                if idx == 0:
                    ci = n_image_input_channels
                items.append((idx, conv(ci, co)))
                kind = "conv" if idx else ("first" if n_image_input_channels <= 4 else "wide")
                plan.append((kind, cname, str(idx), CONV_RELU))
            setattr(self, cname, container(items))
            if bi == 0:
                x1 = len(plan) - 1
            if bi + 1 < len(_ENCODER):
                plan.append(("pool", None, None, 0))
                pooled[bi + 1] = len(plan) - 1

        def skip(src):
            if skip_connections:
                plan.append(("add", None, None, src))

        skip(pooled[4])                                      # decoder_input = x_0_5 + x_0_4_d (models.py:774-777)
        if deconv_decoder:                                   # models.py:618-686
            for cname, ci, co, has_conv, src in [("deconv_0_4", 512, 256, True, pooled[3]), ("deconv_0_3", 256, 128, True, pooled[2]),
                                                 ("deconv_0_2", 128, 64, True, pooled[1]), ("deconv_0_1", 64, 64, False, x1)]:
                items = [(0, deconv(ci, co))]
                plan.append(("deconv", cname, "0", CONV_RELU | CONV_ZEROSTUFF2X))
                if has_conv:
                    items.append((2, conv(co, co)))
                    plan.append(("conv", cname, "2", CONV_RELU))
                setattr(self, cname, container(items))
                skip(src)                                    # models.py:780-799
        else:                                                # models.py:688-733
            self.upsample_0_4 = container([(4, conv(512, 256)), (6, conv(256, 256))])
            self.upsample_0_3 = container([(4, conv(256, 128)), (6, conv(128, 64))])
            plan += [("conv", "upsample_0_4", "4", CONV_RELU | CONV_UPSAMPLE2X), ("conv", "upsample_0_4", "6", 0)]
            skip(pooled[3])                                  # models.py:804-807
            plan += [("conv", "upsample_0_3", "4", CONV_RELU | CONV_UPSAMPLE2X), ("conv", "upsample_0_3", "6", 0)]
            if full_output:
                self.upsample_0_2 = container([(2, conv(64, 64)), (4, conv(64, 64))])
                self.upsample_0_1 = container([(2, conv(64, 64)), (4, conv(64, 64))])
                plan += [("conv", "upsample_0_2", "2", CONV_RELU | CONV_UPSAMPLE2X), ("conv", "upsample_0_2", "4", CONV_RELU),
                         ("conv", "upsample_0_1", "2", CONV_RELU | CONV_UPSAMPLE2X), ("conv", "upsample_0_1", "4", CONV_RELU)]
        self.heads_0 = container([(0, conv(64, 64)), (2, conv(64, 32)), (4, conv(32, n_keypoints))])  # models.py:736-747
        plan += [("conv", "heads_0", "0", CONV_RELU), ("conv", "heads_0", "2", CONV_RELU),
                 ("conv", "heads_0", "4", CONV_OUT_NCHW)]
        if internalize_spatial_softmax:                      # models.py:750-759
            self.softmax = container([(0, SoftArgmaxPavlo(n_keypoints, learned_beta, initial_beta))])
        self._plan = plan
        self._skip_sources = {e[3] for e in plan if e[0] == "add"}
        self._packed = _PackedCache()
        # "fp32": exact fp32 MFMA kernel everywhere.  "fp16x3": inference runs the split-precision kernel
        # (fp32 in/out, 3 fp16 MFMAs per product, fp32-class error); training always uses the fp32 kernels.
        self.precision = "fp32"
        # fp32 3x3 stride-1 convs: "winograd" = F(2x2,3x3) on the fp32 matrix cores wherever it is the faster exact-fp32
        # form (csrc/conv_wino.hip: 2.25x fewer MFMA cycles, same IEEE fp32 arithmetic up to round-off), "direct" = the
        # implicit-GEMM kernel everywhere (csrc/conv_mfma.hip, the reference form)
        self.conv_algorithm = os.environ.get("DREAM_CONV_ALGORITHM", "winograd")
        self.overlap_wgrad = os.environ.get("DREAM_OVERLAP_WGRAD", "1") != "0"
        # training forward: a conv that feeds MaxPool2d(2) stores the pooled tensor from its own epilogue (csrc/conv_wino4.hip MODE 4)
        # instead of a stand-alone max-pool pass over the un-pooled one; "0": the separate pass (A/B, tests)
        self.pool_in_training_conv = os.environ.get("DREAM_POOL_IN_TRAINING_CONV", "1") != "0"
        self._aux = {}
        # the reference builds every hourglass on vgg19(pretrained=True).features (models.py:587): ImageNet weights for all
        # encoder convs but the first when they can be had, one loud warning otherwise (dream_amd/pretrained.py)
        self.imagenet_initialised = _pretrained.init_vgg19_encoder(self)

    # ---- helpers -------------------------------------------------------------------------------------
    def _layer(self, cname, child):
        return getattr(getattr(self, cname), child)

    def plan_layers(self):
        """[(kind, module-or-None, flags)] in execution order."""
        return [(kind, self._layer(c, ch) if c else None, flags) for kind, c, ch, flags in self._plan]

    def _packed_aux(self, mod):
        """[Cin_T,Cout_T,3,3] ConvTranspose weight packed as the stride-2 conv that is its data gradient."""
        key = ("s2", id(mod.weight))
        tag = (mod.weight._version, mod.weight.data_ptr())
        hit = self._aux.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                hit = (tag, ops.pack_conv_weight(mod.weight.detach(), 0))
            self._aux[key] = hit
        return hit[1]

    def plan_parameters(self):
        out = []
        for kind, mod, _ in self.plan_layers():
            if mod is not None:
                out += [mod.weight, mod.bias]
        return out

    def output_resolution(self, input_wh):
        """(W,H) of the belief maps for a (W,H) input -- what the reference learns by pushing a zero
        image through the model (dream/network.py:397-418), computed arithmetically here."""
        w, h = int(input_wh[0]), int(input_wh[1])
        for kind, _, flags in self.plan_layers():
            if kind == "pool":
                w, h = w // 2, h // 2
            elif kind in ("conv", "deconv") and flags & (CONV_UPSAMPLE2X | CONV_ZEROSTUFF2X):
                w, h = w * 2, h * 2
        return (w, h)

    def input_channel_pad(self):
        """Channel count of the NHWC tensor a "wide" first conv reads (fp32 kernel: multiples of 16; split kernel: 32)."""
        return ops.round_up(self.n_image_input_channels, 32 if self.precision == "fp16x3" else 16)

    def _check_input(self, x, x_is_nhwc):
        ok = x.dim() == 4 and (int(x.shape[3]) == self.input_channel_pad() if x_is_nhwc
                               else int(x.shape[1]) == self.n_image_input_channels)
        if not ok:
            raise RuntimeError("expected [B,%d,H,W] input, got %s" % (self.n_image_input_channels, tuple(x.shape)))

    def _fuse_pool(self, layers, li, x_nhwc):
        """Inference: fold the 2x2 max-pool that follows conv ``li`` into its epilogue?  Not when the un-pooled tensor is
        a skip source, and only at >= 160 px: the fused kernel needs even tile sides, which costs 25 % more tiles at
        50x50 (13.8 vs 11.3 + 0.15 ms, profiles/r01_pool_fusion.txt) but saves a 1.3 ms pass at 400x400."""
        if not (li + 1 < len(layers) and layers[li + 1][0] == "pool") or li in self._skip_sources:
            return False
        mod = layers[li][1]
        if (self.conv_algorithm == "winograd" and self.precision == "fp32" and mod is not None and int(mod.weight.shape[1]) == int(x_nhwc.shape[3])
                and ops.winograd_tile(int(x_nhwc.shape[1]), int(x_nhwc.shape[2]), int(x_nhwc.shape[3]), int(mod.weight.shape[0]), int(x_nhwc.shape[0])) == 4):
            return True         # F(4x4,3x3) works on whole 4x4 tiles anyway: the pooled store is free at every map size (saves the 100^2 / 50^2 passes)
        return min(int(x_nhwc.shape[1]), int(x_nhwc.shape[2])) >= 160

    def _first_pair_subbatch(self, layers, li, x_nchw, save):
        """Frames per sub-batch for the first conv pair of an inference pass (0: run the layers whole).  DREAM_FIRST_SUBBATCH=n; only
        where the second conv runs on the F(4x4,3x3) kernel with the pool fused, is no skip source, and the batch has several sub-batches."""
        n = int(os.environ.get("DREAM_FIRST_SUBBATCH", "0"))
        if n <= 0 or save or self.precision != "fp32" or self.conv_algorithm != "winograd" or li + 2 >= len(layers):
            return 0
        (k1, m1, f1), (k2, _, _) = layers[li + 1], layers[li + 2]
        b, _, h, w = (int(v) for v in x_nchw.shape)
        if (k1 != "conv" or k2 != "pool" or f1 != CONV_RELU or any(i in self._skip_sources for i in (li, li + 1, li + 2)) or b <= n
                or h % 2 or w % 2 or int(m1.weight.shape[1]) != int(layers[li][1].weight.shape[0])):
            return 0
        cin, cout = int(m1.weight.shape[1]), int(m1.weight.shape[0])
        if not self._use_winograd(cin, cout, f1) or ops.winograd_tile(h, w, cin, cout, min(n, b)) != 4:
            return 0
        return n

    def _use_winograd(self, cin, cout, flags):
        """Winograd serves the plain 3x3 convs (bias, ReLU, fused max-pool); measured faster than the direct kernel for every
        DREAM layer with >= 64 output channels (profiles/r02_microbench_wino_b128.txt: 1.5-2.05x), on par at 32."""
        if self.conv_algorithm != "winograd":
            if self.conv_algorithm != "direct":
                raise ValueError("unknown conv_algorithm %r" % (self.conv_algorithm,))
            return False
        return cin % 16 == 0 and cin >= 32 and cout >= 64 and not (flags & ~(CONV_RELU | CONV_POOL2 | ops.CONV_RELUMASK))

    def _ups_winograd(self, cin, cout):
        """The convs that follow nn.Upsample(2), as 4x4 stride-2 transposed convs on the Winograd kernel (conv_wino.hip, PAT)."""
        return self.conv_algorithm == "winograd" and cin % 16 == 0 and cin >= 32 and cout > 64

    @staticmethod
    def _join(a, b):
        sa = tuple(a.shape) if torch.is_tensor(a) else tuple(int(v) for v in a)
        if sa != tuple(b.shape):      # what the reference's `+` raises for resolutions the pools do not divide
            raise RuntimeError("The size of tensor a %s must match the size of tensor b %s" % (sa, tuple(b.shape)))

    # ---- execution -------------------------------------------------------------------------------------
    def run_forward_f16x3(self, x, params, x_is_nhwc=False, x_amax=None):
        """Inference plan on the split-precision conv kernel.  Each kernel publishes max|y| of its output (amax side
        channel) so the next conv can scale its input into fp16 range; pooling cannot raise the maximum."""
        act, amax = x, x_amax
        pi = 0
        layers = self.plan_layers()
        keep = {}
        pool_done = False
        for li, (kind, mod, flags) in enumerate(layers):
            if kind == "pool":
                if not pool_done:
                    act = ops.maxpool2(act)
                pool_done = False
            elif kind == "add":
                self._join(act, keep[flags])
                act, amax = ops.add(act, keep[flags], want_amax=True)
            else:
                w, bias = params[pi], params[pi + 1]
                pi += 2
                if kind == "first":
                    act, amax = ops.conv3x3_first_amax(act, w, bias, relu=bool(flags & CONV_RELU))
                else:
                    if kind == "wide" and not x_is_nhwc:
                        amax = ops.absmax(act)
                        act = ops.nchw_to_nhwc(act, cpad=self.input_channel_pad())
                    if self._fuse_pool(layers, li, act):
                        flags, pool_done = flags | CONV_POOL2, True
                    if flags & CONV_UPSAMPLE2X:              # upsample + conv == a 4x4 transposed conv (see run_forward)
                        pk4 = self._packed.get(mod.weight, "ups", f16x3=True)
                        act, amax = ops.conv_transpose4x4s2_f16x3(act, amax, pk4, pk4[3], None, bias, flags & CONV_RELU, direct_taps=36)
                    elif kind == "deconv":
                        p16 = self._packed.get(mod.weight, 1, f16x3=True)
                        act, amax = ops.conv_transpose3x3s2_f16x3(act, amax, p16, p16[3], bias, relu=bool(flags & CONV_RELU))
                    else:
                        p16 = self._packed.get(mod.weight, 0, f16x3=True)
                        act, amax = ops.conv2d_f16x3(act, amax, p16, p16[3], 3, None, bias, None, flags,
                                                     want_amax=not (flags & CONV_OUT_NCHW))
            if li in self._skip_sources:
                keep[li] = act
        return act

    def run_forward(self, x, params, save, x_is_nhwc=False, x_amax=None):
        """Executes the plan.  ``params`` is plan_parameters() (possibly autograd-detached).  With
        ``save`` the per-layer inputs/outputs needed by run_backward are returned as well.  ``x_is_nhwc``: the
        caller (multi-stage) already built the zero-padded NHWC input of a "wide" first conv."""
        self._check_input(x, x_is_nhwc)
        if self.precision == "fp16x3" and not save:
            return self.run_forward_f16x3(x, params, x_is_nhwc, x_amax), []
        if self.precision not in ("fp32", "fp16x3"):
            raise ValueError("unknown precision %r" % (self.precision,))
        saved = []
        keep = {}
        act = x
        pi = 0
        layers = self.plan_layers()
        pool_done = add_done = False
        pooled_by_conv = None
        consumed = 0
        for li, (kind, mod, flags) in enumerate(layers):
            if consumed > 0:                               # this layer ran inside the sub-batched first pair below
                consumed -= 1
                if kind not in ("pool", "add"):
                    pi += 2
                continue
            inp = act
            skip = None
            if kind == "pool":
                if pooled_by_conv is not None:             # training: the conv's own launch stored the pooled tensor beside the un-pooled one
                    act, pooled_by_conv = pooled_by_conv, None
                elif not pool_done:
                    act = ops.maxpool2(inp)
                pool_done = False
            elif kind == "add":
                if not add_done:
                    self._join(inp, keep[flags])
                    act, _ = ops.add(inp, keep[flags])
                add_done = False
            else:
                if not save and kind not in ("first", "wide") and self._fuse_pool(layers, li, inp):
                    flags, pool_done = flags | CONV_POOL2, True
                # inference: the skip connection that follows this conv (models.py:774-799: x = x + x_0_k_d) is added in its own
                # epilogue, after the ReLU -- no separate elementwise launch, the sum never round-trips through HBM.  (Training
                # keeps the separate add: the backward pass reads the conv's own output as its ReLU mask.)
                if (not save and kind in ("conv", "deconv") and li + 1 < len(layers) and layers[li + 1][0] == "add"
                        and not (flags & (CONV_POOL2 | CONV_UPSAMPLE2X | CONV_OUT_NCHW)) and li not in self._skip_sources):
                    skip = keep[layers[li + 1][2]]
                w, bias = params[pi], params[pi + 1]
                pi += 2
                sub = self._first_pair_subbatch(layers, li, inp, save) if kind == "first" else 0
                if sub:
                    # inference: conv1_1 -> conv1_2 (+ pool) over sub-batches of `sub` frames, so that the 64-channel full-resolution tensor
                    # between them (41 MB per frame at 400 x 400: 5.2 GB at 128 frames, written and re-read through HBM) is a
                    # `sub`-frame buffer that is re-used in place and stays in the 256-MB Infinity Cache (dream/models.py:591-599)
                    mod2 = layers[li + 1][1]
                    u2, rows2 = self._packed.get(mod2.weight, "wino4_0")
                    b_all, h_in, w_in = int(inp.shape[0]), int(inp.shape[2]), int(inp.shape[3])
                    act = torch.empty((b_all, h_in // 2, w_in // 2, rows2), dtype=torch.float32, device=inp.device)
                    for s0 in range(0, b_all, sub):
                        a1 = ops.conv3x3_first(inp[s0:s0 + sub], w, bias, relu=bool(flags & CONV_RELU))
                        ops.conv3x3_winograd4(a1, u2, rows2, None, params[pi + 1], None, layers[li + 1][2] | CONV_POOL2, out=act[s0:s0 + sub])
                        del a1
                    consumed = 2                             # conv1_2 and its pool
                elif kind == "first":
                    act = ops.conv3x3_first(inp, w, bias, relu=bool(flags & CONV_RELU))
                else:
                    if kind == "wide" and not x_is_nhwc:
                        inp = ops.nchw_to_nhwc(inp, cpad=self.input_channel_pad())
                    if flags & CONV_UPSAMPLE2X:              # upsample + conv == a 4x4 transposed conv: 4 MACs / output, not 9
                        if self._ups_winograd(int(inp.shape[3]), int(mod.weight.shape[0])):
                            # ... and that transposed conv by minimal filtering on the Winograd kernel: 9/16 of those again
                            tile = ops.convT4x4_winograd_tile(inp, int(mod.weight.shape[0]))
                            u4, cout4 = self._packed.get(mod.weight, "ups_wino4_0" if tile == 4 else "ups_wino0")
                            act = ops.conv_transpose4x4s2_winograd_tile(tile, inp, u4, cout4, None, bias, flags & CONV_RELU, direct_taps=36)
                        else:
                            pk4, cout4 = self._packed.get(mod.weight, "ups")
                            act = ops.conv_transpose4x4s2(inp, pk4, cout4, None, bias, flags & CONV_RELU, direct_taps=36)
                    elif kind == "deconv":                   # ConvTranspose weight [Cin,Cout,3,3], mode-1 packing; sub-pixel
                        packed, rows, _, _ = self._packed.get(mod.weight, 1)   # phases: a quarter of the zero-stuffed MACs
                        if skip is not None:
                            self._join((inp.shape[0], 2 * inp.shape[1], 2 * inp.shape[2], rows), skip)
                        act = ops.conv_transpose3x3s2(inp, packed, bias, rows, relu=bool(flags & CONV_RELU), skip=skip)
                        add_done = skip is not None
                    elif int(mod.weight.shape[1]) == int(inp.shape[3]) and self._use_winograd(int(inp.shape[3]), int(mod.weight.shape[0]), flags):
                        tile = ops.winograd_tile(int(inp.shape[1]), int(inp.shape[2]), int(inp.shape[3]), int(mod.weight.shape[0]), int(inp.shape[0]))
                        u, rows = self._packed.get(mod.weight, "wino4_0" if tile == 4 else "wino0")
                        if skip is not None:
                            self._join(tuple(inp.shape[:3]) + (rows,), skip)
                        if (save and tile == 4 and skip is None and self.pool_in_training_conv and flags == CONV_RELU
                                and li + 1 < len(layers) and layers[li + 1][0] == "pool"):
                            # training: un-pooled tensor (kept for the backward pass) AND pooled tensor from one launch
                            act, pooled_by_conv = ops.conv3x3_winograd4_pool_both(inp, u, rows, bias, flags)
                        else:
                            act = ops.conv3x3_winograd_tile(tile, inp, u, rows, None, bias, skip, flags | (ops.CONV_RES_AFTER_RELU if skip is not None else 0))
                        add_done = skip is not None
                    elif skip is not None:
                        packed, rows, _, _ = self._packed.get(mod.weight, 0)
                        self._join(tuple(inp.shape[:3]) + (rows,), skip)
                        act = ops.conv2d(inp, packed, rows, 3, 1, None, bias, skip, flags | ops.CONV_RES_AFTER_RELU)
                        add_done = True
                    else:
                        packed, rows, _, _ = self._packed.get(mod.weight, 0)
                        act = ops.conv3x3(inp, packed, bias, rows, flags)
            if save:
                saved.append((inp, act))
            if li in self._skip_sources:
                keep[li] = act
        return act, saved

    def run_backward(self, saved, grad_out_nchw, need_input_grad=False, reducer=None):
        """dL/d(belief maps) [B,K,Ho,Wo] -> list of parameter gradients in plan_parameters() order (and, for a
        multi-stage hourglass, dL/d(NHWC input of the "wide" first conv)).  ``reducer``: overlapped data-parallel
        all-reduce that is fed every gradient as soon as it exists."""
        layers = self.plan_layers()
        grads = _GradList(2 * sum(1 for k, m, _ in layers if m is not None), reducer)
        # weight gradients are leaves of the data-gradient chain: second stream when the batch is small (see _SideStream)
        sh = saved[0][0].shape                             # NCHW image ("first") or NHWC packed input ("wide")
        input_px = int(sh[0]) * (int(sh[2]) * int(sh[3]) if layers[0][0] == "first" else int(sh[1]) * int(sh[2]))
        side = _SideStream.create(grad_out_nchw, self.overlap_wgrad and input_px <= self.OVERLAP_MAX_PIXELS)
        pi = len(grads)
        g = None
        g_input = None
        pending = {}                                       # skip source plan index -> gradient that branched off
        masked = False                                     # g already carries the ReLU gradient of layer li

        def relu_feeds(idx):
            """Entry idx is a ReLU conv whose output is consumed ONLY by entry idx+1: its ReLU gradient can be applied
            by whatever produces d/d(output) (a data-gradient conv epilogue or the max-pool backward)."""
            return (idx >= 0 and layers[idx][0] in ("first", "wide", "conv", "deconv") and bool(layers[idx][2] & CONV_RELU)
                    and idx not in self._skip_sources)

        early = getattr(reducer, "early_marker", None)    # single-process exchange: the plan entry from which the early bucket is final
        for li in range(len(layers) - 1, -1, -1):
            kind, mod, flags = layers[li]
            inp, out = saved[li]
            if early is not None and li < early:           # every entry >= early is done: pack the early bucket (a leaf), mark it
                k_early = reducer.early_k
                _early_bucket_hook(reducer, early, side, lambda: list(grads[k_early:]))
                early = None
            if li in pending:                              # two consumers of this activation: gradients add
                g = ops.add_(g, pending.pop(li))
            if kind == "pool":
                masked = relu_feeds(li - 1)
                g = ops.maxpool2_bwd(g, inp, relu=masked)
                continue
            if kind == "add":
                pending[flags] = ops.clone(g)              # later in-place ReLU masks must not touch this copy
                masked = False
                continue
            pi -= 2
            fuse = relu_feeds(li - 1)                      # this layer's data gradient can carry layer li-1's ReLU mask
            if kind == "deconv":
                # ConvTranspose2d(3,2,1,op 1) + ReLU (models.py:621-686): bias grad = column sums, weight grad over the
                # stride-2 taps of dy, data grad = the 3x3 stride-2 conv of dy with the (un-flipped) weight
                if not masked:
                    g = ops.relu_bwd_(g, out)
                def leaf(pi=pi, inp=inp, g=g):
                    grads[pi] = ops.convT_wgrad(inp, g, 3)
                    grads[pi + 1] = ops.channel_sum(g)
                _on_side(side, leaf, inp, g)
                packed_s2, rows_s2, _ = self._packed_aux(mod)
                g = ops.conv2d(g, packed_s2, rows_s2, 3, 2, None, None, inp if fuse else None,
                               ops.CONV_RELUMASK if fuse else 0)
                masked = fuse
                continue
            cout, cin = int(mod.weight.shape[0]), int(mod.weight.shape[1])
            if flags & CONV_OUT_NCHW:
                g = ops.nchw_to_nhwc(grad_out_nchw, cpad=ops.round_up(cout, 16))   # zero-padded K -> 16k channels
            if flags & CONV_RELU and not masked:
                g = ops.relu_bwd_(g, out)
            masked = False
            if kind == "first":
                grads[pi], grads[pi + 1] = ops.conv3x3_first_wgrad(inp, g)
                g = None
                continue
            if kind == "wide":
                dw, db = ops.conv3x3_wgrad(inp, g, cout, int(inp.shape[3]), 0)
                grads[pi], grads[pi + 1] = dw[:, :cin].contiguous(), db
                if need_input_grad:
                    packed_t, rows, _, _ = self._packed.get(mod.weight, 1)
                    g_input = ops.conv3x3(g, packed_t, None, rows, 0)         # [B,H,W,cin]
                g = None
                continue
            def leaf(pi=pi, inp=inp, g=g, cout=cout, cin=cin, ups=flags & CONV_UPSAMPLE2X):
                if (self.conv_algorithm == "winograd" and int(inp.shape[3]) == cin
                        and ops.wgrad_winograd_pays(int(g.shape[0]) * int(g.shape[1]) * int(g.shape[2]), cin, cout)
                        and (not ups or (cin % 64 == 0 and cout % 64 == 0 and g.shape[1] % 2 == 0 and g.shape[2] % 2 == 0))):
                    # 16/36 of the multiplications; the nearest x2 upsample in front of the conv is fused into the patch load
                    grads[pi], grads[pi + 1] = ops.conv3x3_wgrad_winograd(inp, g, cout, cin, flags=CONV_UPSAMPLE2X if ups else 0)
                else:
                    grads[pi], grads[pi + 1] = ops.conv3x3_wgrad(inp, g, cout, cin, ups)
            _on_side(side, leaf, inp, g)
            packed_t, rows, _, cols_pad = self._packed.get(mod.weight, 1)
            if int(g.shape[3]) != cols_pad:
                raise RuntimeError("internal: gradient has %d channels, packed weights expect %d" % (g.shape[3], cols_pad))
            if flags & CONV_UPSAMPLE2X:                    # the mask lives at half resolution: after upsample2_bwd
                if (self._ups_winograd(cin, cout) and cin > 64 and int(g.shape[3]) == cout
                        and g.shape[1] % 2 == 0 and g.shape[2] % 2 == 0):
                    # data gradient of the equivalent transposed conv, straight at half resolution (no full-resolution
                    # intermediate, no upsample2_bwd pass): four phase convs of nine positions each
                    tile = ops.conv4x4s2_winograd_tile_of(g, cin)
                    u4b, rows_b = self._packed.get(mod.weight, "ups_wino4_1" if tile == 4 else "ups_wino1")
                    g = ops.conv4x4s2_winograd_tile(tile, g, u4b, rows_b)
                elif self._use_winograd(int(g.shape[3]), cin, 0) and int(g.shape[3]) == cout:
                    tile = ops.winograd_tile(int(g.shape[1]), int(g.shape[2]), cout, cin, int(g.shape[0]))
                    u_t, rows_t = self._packed.get(mod.weight, "wino4_1" if tile == 4 else "wino1")
                    g = ops.upsample2_bwd(ops.conv3x3_winograd_tile(tile, g, u_t, rows_t, None, None, None, 0))
                else:
                    g = ops.upsample2_bwd(ops.conv3x3(g, packed_t, None, rows, 0))
            elif self._use_winograd(int(g.shape[3]), cin, ops.CONV_RELUMASK if fuse else 0) and int(g.shape[3]) == cout:
                tile = ops.winograd_tile(int(g.shape[1]), int(g.shape[2]), cout, cin, int(g.shape[0]))
                u_t, rows_t = self._packed.get(mod.weight, "wino4_1" if tile == 4 else "wino1")   # data gradient = conv with the transposed, flipped taps
                g = ops.conv3x3_winograd_tile(tile, g, u_t, rows_t, None, None, inp if fuse else None, ops.CONV_RELUMASK if fuse else 0)
                masked = fuse
            else:
                g = ops.conv3x3(g, packed_t, None, rows, 0, relu_mask=inp if fuse else None)
                masked = fuse
        if side is not None:
            side.join()
        if need_input_grad:
            return grads, g_input
        return grads

    # ---- data-parallel interface (dream_amd/data_parallel.py): one replica's share of a step ------------------------
    def dp_parameters(self):
        """Parameters in the order dp_backward returns their gradients."""
        return self.plan_parameters()

    def dp_early_bucket(self):
        """-> (index k into dp_parameters(), plan entry li): the gradients of dp_parameters()[k:] are final once the backward plan has
        finished the entries >= li.  The late bucket is the longest prefix of the plan that holds at most 4 % of the parameters -- for
        vgg_q conv1_1 .. conv3_1, whose data / weight gradients at 400 x 400 and 200 x 200 are a quarter of the backward pass."""
        layers = self.plan_layers()
        total = sum(m.weight.numel() + m.bias.numel() for _, m, _ in layers if m is not None)
        seen, k = 0, 0
        for li, (_, m, _) in enumerate(layers):
            if m is None:
                continue
            n = m.weight.numel() + m.bias.numel()
            if seen + n > 0.04 * total:
                return (k, li) if k > 0 else None
            seen, k = seen + n, k + 2
        return None

    def dp_trainable(self):
        return True

    def dp_forward(self, x, save):
        """-> ([differentiable outputs], context for dp_backward)."""
        out, saved = self.run_forward(x, [p.detach() for p in self.plan_parameters()], save=save)
        return [out], saved

    def dp_backward(self, saved, grad_outs, reducer=None):
        return list(self.run_backward(saved, grad_outs[0].contiguous(), reducer=reducer))

    def dp_finish(self, outs):
        """Gathered differentiable outputs -> what forward() returns (the soft-argmax head is computed on the gathered maps)."""
        return outs + [self.softmax[0](outs[0])] if self.internalize_spatial_softmax else outs

    def forward(self, x):
        params = self.plan_parameters()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            out = _HourglassFunction.apply(self, x, *params)
        else:
            with torch.no_grad():
                out, _ = self.run_forward(x, [p.detach() for p in params], save=False)
        outputs = [out]
        if self.internalize_spatial_softmax:
            outputs.append(self.softmax[0](out))
        return outputs


class _HourglassFunction(torch.autograd.Function):
    """Whole-network autograd node: forward keeps the NHWC activations, backward runs the HIP
    backward plan and (when torch.distributed is initialised, one process per GPU) sums the flat
    gradient buffer over ranks with ONE RCCL all-reduce before handing the views to autograd."""

    @staticmethod
    def forward(ctx, module, x, *params):
        out, saved = module.run_forward(x.detach(), [p.detach() for p in params], save=True)
        ctx.module = module
        ctx.saved_acts = saved
        return out

    @staticmethod
    def backward(ctx, grad_out):
        module = ctx.module
        reducer = _OverlappedAllReduce.create()
        grads = _guarded_backward(module.run_backward, ctx.saved_acts, grad_out.contiguous(), reducer=reducer)
        ctx.saved_acts = None
        return (None, None) + tuple(_reduced(reducer, grads))


def allreduce_gradients(grads):
    """Data-parallel exchange step.  Every rank holds an equal chunk of the global
    batch and has the gradient of its LOCAL mean loss; the gradient of the global mean loss is their
    average, so: one all-reduce(sum) over a single flat fp32 buffer (RCCL over xGMI on GPUs, gloo in
    the CPU tests), then a scale by 1/world_size.  No-op in single-process runs."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return grads
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    out, o = [], 0
    for g in grads:
        n = g.numel()
        out.append(flat[o:o + n].view_as(g))
        o += n
    return out


class _OverlappedAllReduce:
    """Data-parallel exchange overlapped with backward.  Gradients are handed over in the order backward produces them
    (identical on every rank); every ~32 MB they are flattened into one buffer whose all-reduce(sum) is started
    asynchronously (RCCL runs it on its own stream over xGMI while the remaining layers' backward kernels run);
    ``finish()`` waits, scales by 1/world and maps each gradient to its slice.  ResNet-101 at 16 frames per GPU moves
    216 MB per step: exposed at the end of backward that is ~6 % of the step, overlapped it disappears."""
    BUCKET_BYTES = 32 << 20

    @staticmethod
    def create():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("DREAM_FORCE_REDUCER")):
            return _OverlappedAllReduce(dist)
        return None

    def __init__(self, dist):
        self.dist, self.world = dist, dist.get_world_size()
        self.current, self.current_bytes, self.inflight, self.views, self.producers = [], 0, [], {}, set()

    def add(self, g):
        self.current.append(g)
        if g.is_cuda:
            self.producers.add(torch.cuda.current_stream())
        self.current_bytes += g.numel() * g.element_size()
        if self.current_bytes >= self.BUCKET_BYTES:
            self._flush()

    def _flush(self):
        if not self.current:
            return
        if self.producers:                                  # gradients may come from the weight-gradient side stream
            here = torch.cuda.current_stream()
            for st in self.producers:
                if st != here:
                    here.wait_stream(st)
            self.producers = set()
        flat = torch.cat([g.reshape(-1) for g in self.current])
        work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, async_op=True)
        self.inflight.append((work, flat, self.current))
        self.current, self.current_bytes = [], 0

    def finish(self):
        self._flush()
        for work, flat, items in self.inflight:
            work.wait()
            flat /= self.world
            o = 0
            for g in items:
                n = g.numel()
                self.views[id(g)] = flat[o:o + n].view_as(g)
                o += n
        self.inflight = []

    def result(self, g):
        return self.views.get(id(g), g)


class _SideStream:
    """Runs weight-gradient kernels on a second HIP stream.  In backward the data-gradient chain (conv -> BN -> conv ..)
    is the critical path and every weight gradient is a leaf hanging off it; at small per-GPU batches neither kind of
    kernel fills 256 CUs (25x25 / 13x13 feature maps), so the leaves run concurrently with the chain instead of in it.
    ``run(fn, *inputs)``: fn's launches wait for everything queued on the main stream so far; ``join()`` makes the main
    stream wait for the leaves.  Inputs are record_stream()-ed so the caching allocator does not recycle them early."""
    _streams = {}

    @classmethod
    def create(cls, like, enabled=True):
        if not (enabled and like.is_cuda):
            return None
        # Under hipGraph capture (the replicas of the single-process data-parallel path, DREAM_TRAIN_GRAPH) the side branch is captured
        # from a NORMAL-priority stream: a low-priority one captured into the graph slows the replayed step from 344 to 264 frames/s
        # (resnet_h, 16 frames; profiles/r05_ab_train_graph.txt) -- the eager step is the one that gains from the low priority.
        capturing = torch.cuda.is_current_stream_capturing()
        ctl = getattr(_split_capture, "ctl", None)
        if capturing and ctl is not None:
            return _DeferredSide(ctl)                  # a backward captured in segments (below): the leaves get graphs of their own
        if capturing and cls.low_priority_allowed:
            cls.forbid_low_priority()                  # a model that captures was not announced (DreamDataParallel.use_graphs does)
        key = (like.device.index, capturing)
        if key not in cls._streams:
            from . import _hip
            cls._streams[key] = _hip.own_stream(like.device, "capture-leaves") if capturing else cls._new_stream(like.device)
        return cls(cls._streams[key])

    @classmethod
    def live_stream(cls, device):
        """The second stream of eager steps on ``device`` (created on first use): what a backward replayed as a sequence of graphs
        sends its leaf segments to -- the stream that is known to run beside the main stream, not one more hardware queue."""
        key = (device.index, False)
        if key not in cls._streams:
            cls._streams[key] = cls._new_stream(device)
        return cls._streams[key]

    _memory = {}

    @classmethod
    def _device_memory(cls, device):
        """Total memory of ``device`` (cached; a replica's pool thread may not be able to query the device properties: 32 GB then)."""
        idx = device.index if device.index is not None else -1
        if idx not in cls._memory:
            try:
                cls._memory[idx] = int(torch.cuda.get_device_properties(device).total_memory)
            except (AssertionError, RuntimeError):
                return 32 << 30
        return cls._memory[idx]

    low_priority_allowed = True

    @classmethod
    def forbid_low_priority(cls):
        """Processes that replay training steps as hipGraphs get normal-priority weight-gradient streams throughout: with a low-priority
        stream merely PRESENT in the process (the eager warm-up steps create it) the replayed step runs at 259-263 frames/s instead of 344
        (resnet_h, 16 frames; profiles/r05_ab_train_graph.txt), whichever stream the side branch was captured from."""
        if cls.low_priority_allowed:
            cls.low_priority_allowed = False
            cls._streams.clear()

    @staticmethod
    def _new_stream(device):
        """The device's weight-gradient stream: one of the process's own HIP streams (_hip.own_stream -- not one of torch's 32 pooled
        streams, which long-running processes see handed out twice).  DREAM_SIDE_STREAM_PRIORITY=low (opt-in): at the LOWEST HIP stream
        priority, so that the dispatcher prefers the main stream's dependent chain -- the step's critical path -- wherever both have
        workgroups pending; "high": the opposite (A/B); default: normal priority."""
        # "default" is the default: the low priority measured +0.4 % at 16 frames (noise level) and makes the overlap at 128 frames better but
        # still unreliable, while a low-priority stream in the process slows every replayed training GRAPH by a quarter (below) -- and
        # nothing is known about its interplay with RCCL's streams on a multi-GPU node.  Opt-in: DREAM_SIDE_STREAM_PRIORITY=low.
        # Measured round 5 (profiles/r05_ab_side_stream_priority.txt, resnet_h training, one box, alternating): 16 frames 356.7 (normal) /
        # 358.0 (low) / 353.7 (high) frames/s; at 128 frames with the overlap forced the low-priority stream reaches 446-450 against the
        # in-order 434-436 -- but two early runs dropped to 300 as in round 4 and the final measurement run to 414: the threshold stays 96.
        want = os.environ.get("DREAM_SIDE_STREAM_PRIORITY", "default")
        if want == "low" and not _SideStream.low_priority_allowed:
            want = "default"
        from . import _hip
        return _hip.own_stream(device, "leaves", {"high": -1, "low": 1}.get(want, 0))

    def __init__(self, stream):
        self.side, self.main = stream, torch.cuda.current_stream()
        # DREAM_SIDE_KEEP=1: the leaves' inputs are kept referenced until join() instead of record_stream()-ed.  A recorded block cannot be
        # reused by the main stream until an event on the side stream has passed; holding the references until the main stream has
        # been ordered behind the side stream needs no events at all (at the price of the inputs' memory until the end of backward).
        # Measured round 5 (profiles/r05_ab_side_stream_keep.txt, alternating on one box): resnet_h training at 16 frames 359.2 / 359.9 / 360.3
        # (record_stream) -> 364.5 / 364.9 / 363.8 frames/s; at 128 frames equal (451-453).  "0" restores record_stream().
        self.keep = [] if os.environ.get("DREAM_SIDE_KEEP", "1") == "1" else None
        # ... bounded: past DREAM_SIDE_KEEP_MAX_MB (default: an eighth of the device's memory) of retained inputs the remaining leaves fall
        # back to record_stream() -- a batch that used to fit must not run out of memory because of this switch (round-5 advice)
        self.kept_bytes = 0
        self.keep_limit = int(os.environ.get("DREAM_SIDE_KEEP_MAX_MB", "0")) << 20
        if self.keep is not None and self.keep_limit <= 0:
            self.keep_limit = _SideStream._device_memory(stream.device) // 8
        self.batch = int(os.environ.get("DREAM_SIDE_BATCH", "1"))
        self.pending = []
        self.count = 0

    # leaves between create() and join() of the last eager backward OF THIS THREAD (for _DeferredSide: the replicas of the
    # single-process multi-GPU path run their backward passes on one host thread each)
    _counts = threading.local()

    @classmethod
    def leaves_of_last_backward(cls):
        return getattr(cls._counts, "leaves", 0)

    def run(self, fn, *inputs):
        self.count += 1
        if self.batch > 1 and self.keep is not None:
            # leaves are launched in groups: ONE event (record on the main stream, wait on the side stream) per `batch` leaves instead
            # of one per leaf -- the side stream has slack, the main stream's queue carries ~330 fewer barrier packets per ResNet step
            self.pending.append(fn)
            self.keep.append(inputs)
            if len(self.pending) >= self.batch:
                self.flush()
            return None
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side), ops.wgrad_width(ops.SIDE_WGRAD_WIDTH):
            out = fn()
        nbytes = sum(t.numel() * t.element_size() for t in inputs)
        if self.keep is not None and self.kept_bytes + nbytes <= self.keep_limit:
            self.keep.append(inputs)
            self.kept_bytes += nbytes
        else:
            for t in inputs:
                t.record_stream(self.side)
        return out

    def checkpoint(self, cb):
        """Everything queued so far is launched, then ``cb(stream)`` runs with the second stream (an event recorded there is behind every
        leaf queued before this call)."""
        self.flush()
        cb(self.side)

    def flush(self):
        if self.pending:
            self.side.wait_stream(self.main)
            with torch.cuda.stream(self.side), ops.wgrad_width(ops.SIDE_WGRAD_WIDTH):
                for fn in self.pending:
                    fn()
            del self.pending[:]

    def join(self):
        self.flush()
        self.main.wait_stream(self.side)
        _SideStream._counts.leaves = self.count
        if self.keep is not None:
            del self.keep[:]
            self.kept_bytes = 0


_split_capture = threading.local()       # .ctl: the controller of a backward that is being captured in segments (data_parallel._SplitCapture)


class _DeferredSide:
    """The weight-gradient leaves of a backward captured as a SEQUENCE of hipGraphs (DREAM_TRAIN_GRAPH_SPLIT=n, data_parallel
    ._SplitCapture).  One hipGraph with a forked branch replays no faster than the in-order graph (resnet_h, 16 frames: 343 frames/s
    with one, two or four executor queues, profiles/r05_ab_train_graph.txt) while the eager step gains 7 % from its second
    stream.  So the leaves are not forked inside the capture: every ``n`` of them the main capture is cut, the collected leaves
    are captured into a graph of their own (second stream, second memory pool), and the main capture resumes.  Replayed, the
    main segments go to the main stream, the leaf segments to a live second stream ordered behind the segment that produced
    their inputs by an ordinary event: two hardware queues, as in the eager step.  The leaves' inputs stay referenced until
    join(), so no later main segment can be handed their memory while a leaf segment may still read it."""

    def __init__(self, ctl):
        self.ctl, self.pending, self.keep = ctl, [], []
        # The leaves after the last cut run when the main chain has ended: nothing hides them.  With the number of leaves known (the
        # eager step before the capture counted them) the last segments are tapered -- half of what remains, down to single leaves --
        # so that the exposed tail is one leaf, as in the eager step.  DREAM_TRAIN_GRAPH_SPLIT_TAPER=0: equal segments throughout.
        taper = os.environ.get("DREAM_TRAIN_GRAPH_SPLIT_TAPER", "1") == "1"
        self.total, self.seen = (_SideStream.leaves_of_last_backward() if taper else 0), 0

    def run(self, fn, *inputs):
        self.pending.append(fn)
        self.keep.append(inputs)
        self.seen += 1
        want = self.ctl.leaves
        if self.total > self.seen:
            want = max(1, min(want, (self.total - self.seen + len(self.pending)) // 2))
        if len(self.pending) >= want:
            self.ctl.cut(self.pending)
            del self.pending[:]

    def checkpoint(self, cb):
        """The leaves collected so far become a segment of their own; when it has been replayed, ``cb(stream of the leaf segments)`` runs
        (every replay: a plan entry, data_parallel._SplitCapture)."""
        self.ctl.cut(self.pending, after=cb)
        del self.pending[:]

    def join(self):
        self.ctl.cut(self.pending, join=True)
        del self.pending[:]
        del self.keep[:]


def _guarded_backward(fn, *args, **kwargs):
    """Runs a backward plan.  If it raises, the device is synchronised BEFORE the exception leaves this frame: the plan's second stream
    (_SideStream, DREAM_SIDE_KEEP) may still be reading inputs whose only references are the plan's locals -- the traceback keeps those
    alive exactly until the handler below has waited for the kernels (round-5 advice)."""
    try:
        return fn(*args, **kwargs)
    except BaseException:
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()
        raise


def _early_bucket_hook(reducer, marker, side, early_grads):
    """Single-process data parallelism (data_parallel._EarlyBucket): when a backward plan passes ``marker`` -- from here on no gradient
    of the early bucket (the parameters registered from the marker's layer on: for ResNet-101 everything from layer3 up, 97 % of the
    bytes) changes any more -- the bucket is packed into the replica's flat gradient buffer by a leaf on the second stream (behind
    the weight-gradient leaves that produce it) and an event is recorded behind that leaf; the exchange of the bucket waits for the
    event only and overlaps the rest of the backward pass (dream/network.py:244-256,335: nn.DataParallel's gradient reduction)."""
    if reducer is None or getattr(reducer, "early_marker", None) != marker:
        return
    _on_side(side, lambda: reducer.pack_early(early_grads()))
    if side is not None:
        side.checkpoint(reducer.mark_early)
    else:
        reducer.mark_early(torch.cuda.current_stream() if torch.cuda.is_available() else None)


def _on_side(side, fn, *inputs):
    return fn() if side is None else side.run(fn, *inputs)


class _GradList(list):
    """Gradient slots of a backward plan; every assignment is also handed to the overlapped all-reduce (if any)."""

    def __init__(self, n, reducer=None):
        super().__init__([None] * n)
        self._reducer = reducer

    def __setitem__(self, i, g):
        super().__setitem__(i, g)
        if self._reducer is not None and g is not None:
            self._reducer.add(g)


class _GradDict(dict):
    def __init__(self, reducer=None):
        super().__init__()
        self._reducer = reducer

    def __setitem__(self, k, g):
        super().__setitem__(k, g)
        if self._reducer is not None and g is not None:
            self._reducer.add(g)


def _reduced(reducer, grads):
    if reducer is None:
        return list(grads)
    reducer.finish()
    return [reducer.result(g) for g in grads]


class DreamHourglassMultiStage(nn.Module):
    """The reference's DreamHourglassMultiStage (models.py:350-553): ``stage1`` .. ``stageS`` hourglasses (same
    ``state_dict()`` keys); stage s > 1 reads cat([image, belief maps of stage s-1]) -- the maps nearest-upsampled x4
    unless the decoder already reaches input resolution (:487-493).  Here the concatenation, the upsampling and the
    NCHW->NHWC conversion are one kernel (dream_stage_input_nhwc_f32) writing the zero-padded NHWC tensor the MFMA first
    conv of the stage reads; the whole S-stage network is one autograd node."""

    def __init__(self, n_keypoints, n_image_input_channels=3, internalize_spatial_softmax=True, learned_beta=True,
                 initial_beta=1.0, n_stages=2, skip_connections=False, deconv_decoder=False, full_output=False):
        super().__init__()
        self.n_keypoints = n_keypoints
        self.n_image_input_channels = n_image_input_channels
        self.internalize_spatial_softmax = internalize_spatial_softmax
        self.skip_connections = skip_connections
        self.deconv_decoder = deconv_decoder
        self.full_output = full_output
        if internalize_spatial_softmax:
            # models.py:373-378: the soft-argmax head of every stage is built (it owns parameters) but never returned
            print("WARNING: Keypoint softmax output head is currently unused. Prefer training new models of this type "
                  "with internalize_spatial_softmax = False.")
            self.n_output_heads = 2
            self.learned_beta = learned_beta
            self.initial_beta = initial_beta
        else:
            self.n_output_heads = 1
            self.learned_beta = False
        assert isinstance(n_stages, int), 'Expected "n_stages" to be an integer, but it is {}.'.format(type(n_stages))
        assert 0 < n_stages and n_stages <= 6, \
            "DreamHourglassMultiStage can only be constructed with 1 to 6 stages at this time."
        self.num_stages = n_stages
        for s in range(1, n_stages + 1):
            cin = n_image_input_channels + (n_keypoints if s > 1 else 0)
            setattr(self, "stage%d" % s, DreamHourglass(
                n_keypoints, cin, internalize_spatial_softmax, learned_beta, initial_beta,
                skip_connections=skip_connections, deconv_decoder=deconv_decoder, full_output=full_output))

    @property
    def precision(self):
        return self.stage1.precision

    @precision.setter
    def precision(self, value):
        for st in self.stages():
            st.precision = value

    @property
    def conv_algorithm(self):
        return self.stage1.conv_algorithm

    @conv_algorithm.setter
    def conv_algorithm(self, value):
        for st in self.stages():
            st.conv_algorithm = value

    def stages(self):
        return [getattr(self, "stage%d" % s) for s in range(1, self.num_stages + 1)]

    def map_upsampling(self):
        return 1 if (self.deconv_decoder or self.full_output) else 4

    def output_resolution(self, input_wh):
        return self.stage1.output_resolution(input_wh)

    def plan_parameters(self):
        return [p for st in self.stages() for p in st.plan_parameters()]

    def run_forward(self, x, stage_params, save):
        """-> ([maps of stage 1..S], per-stage saved activations)."""
        if x.dim() != 4 or int(x.shape[1]) != self.n_image_input_channels:
            raise RuntimeError("expected [B,%d,H,W] input, got %s" % (self.n_image_input_channels, tuple(x.shape)))
        outs, saved = [], []
        up = self.map_upsampling()
        for s, (st, params) in enumerate(zip(self.stages(), stage_params)):
            if s == 0:
                y, sv = st.run_forward(x, params, save)
            else:
                prev = outs[-1]
                if (int(prev.shape[2]) * up, int(prev.shape[3]) * up) != (int(x.shape[2]), int(x.shape[3])):
                    # what torch.cat raises in the reference when the pools do not divide the input resolution
                    raise RuntimeError("Sizes of tensors must match except in dimension 1. Expected size %d but got size %d"
                                       % (int(x.shape[2]), int(prev.shape[2]) * up))
                want_amax = st.precision == "fp16x3" and not save
                inp, amax = ops.stage_input(x, prev, up, st.input_channel_pad(), want_amax=want_amax)
                y, sv = st.run_forward(inp, params, save, x_is_nhwc=True, x_amax=amax)
            outs.append(y)
            saved.append(sv)
        return outs, saved

    def run_backward(self, saved, grad_outs, reducer=None):
        """grad_outs: dL/d(maps of stage s) or None, s = 1..S -> parameter gradients in plan_parameters() order."""
        stages = self.stages()
        up = self.map_upsampling()
        ci, k = self.n_image_input_channels, self.n_keypoints
        per_stage = [None] * len(stages)
        carry = None                                       # dL/d(maps of stage s) through stage s+1's input
        for s in range(len(stages) - 1, -1, -1):
            g = grad_outs[s]
            if g is None and carry is None:
                per_stage[s] = [torch.zeros_like(p) for p in stages[s].plan_parameters()]
                continue
            if g is None:
                g = carry
            elif carry is not None:
                g = ops.add(g.contiguous(), carry)[0]
            if s > 0:
                per_stage[s], g_in = stages[s].run_backward(saved[s], g.contiguous(), need_input_grad=True, reducer=reducer)
                b, h, w, c = (int(v) for v in g_in.shape)
                carry = ops.stage_input_bwd(g_in, ci, k, up)
            else:
                per_stage[s] = stages[s].run_backward(saved[s], g.contiguous(), reducer=reducer)
        return [g for grads in per_stage for g in grads]

    def dp_parameters(self):
        return self.plan_parameters()

    def dp_trainable(self):
        return True

    def dp_forward(self, x, save):
        return self.run_forward(x, [[p.detach() for p in st.plan_parameters()] for st in self.stages()], save)

    def dp_backward(self, saved, grad_outs, reducer=None):
        return self.run_backward(saved, [None if g is None else g.contiguous() for g in grad_outs], reducer=reducer)

    def dp_finish(self, outs):
        return outs

    def forward(self, x, verbose=False):
        stage_params = [st.plan_parameters() for st in self.stages()]
        flat = [p for ps in stage_params for p in ps]
        if torch.is_grad_enabled() and any(p.requires_grad for p in flat):
            return list(_MultiStageFunction.apply(self, x, *flat))
        with torch.no_grad():
            outs, _ = self.run_forward(x, [[p.detach() for p in ps] for ps in stage_params], save=False)
        return outs


class _MultiStageFunction(torch.autograd.Function):
    """All S hourglasses as one autograd node with S outputs (same exchange step as _HourglassFunction)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        stage_params, o = [], 0
        for st in module.stages():
            n = len(st.plan_parameters())
            stage_params.append([p.detach() for p in params[o:o + n]])
            o += n
        outs, saved = module.run_forward(x.detach(), stage_params, save=True)
        ctx.module = module
        ctx.saved_acts = saved
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        module = ctx.module
        reducer = _OverlappedAllReduce.create()
        grads = _guarded_backward(module.run_backward, ctx.saved_acts, list(grad_outs), reducer=reducer)
        ctx.saved_acts = None
        return (None, None) + tuple(_reduced(reducer, grads))


class ResnetSimple(nn.Module):
    """The reference's ResnetSimple (models.py:17-155): torchvision ResNet101 trunk + 4 (or 5) x
    [ConvTranspose2d(4,2,1) -> BatchNorm -> ReLU] + 1x1 conv to K maps; identical ``state_dict()``.
    Evaluation mode runs entirely on the MFMA conv kernel: 1x1 / 3x3 / strided convs with the eval-mode
    BatchNorm folded into the epilogue (y = conv*scale + shift), the Bottleneck residual add + ReLU fused
    into the third conv, the 7x7 stem as im2col + 1-tap conv, the 4x4 transposed convs by sub-pixel
    decomposition.  Training mode uses batch-statistics BatchNorm kernels (csrc/bn.hip) between the convs and
    a tape-driven backward plan (data gradients, weight gradients, BN gradients) built from the same kernels."""

    def __init__(self, n_keypoints=7, freeze=False, pretrained=True, full=False):
        super().__init__()
        self.full = full
        self.n_keypoints = n_keypoints
        self._cache = {}
        self.precision = "fp32"        # "fp16x3": evaluation-mode forward on the split-precision conv kernel
        self.conv_algorithm = os.environ.get("DREAM_CONV_ALGORITHM", "winograd")   # see DreamHourglass.conv_algorithm
        # stride-1 1x1 convs (forward and data gradient): "gemm" = the LDS-free GEMM kernel (gemm1x1.hip), "direct" = conv_mfma
        self.conv1x1_algorithm = os.environ.get("DREAM_CONV1X1_ALGORITHM", "gemm")
        # decoder ConvTranspose2d(k4,s2,p1) forward: "winograd" = minimal filtering on the Winograd kernel (9/16 of the direct
        # multiplications, conv_wino.hip), "direct" = sub-pixel phases on conv_mfma
        self.convT_algorithm = os.environ.get("DREAM_CONVT_ALGORITHM", "winograd")
        # weight gradients on a second stream, concurrent with the data-gradient chain (DREAM_OVERLAP_WGRAD=0: in order), up to
        # overlap_max_frames 400x400 frames per step (beyond, each kernel fills the chip on its own).  Round 5 tried 128: with the
        # weight-gradient stream at the lowest HIP priority the overlap gave 447-453 frames/s at 128 frames in twelve runs on four boxes
        # (in order: 435-438) -- and 414 in the round's final measurement run, 300 in two early ones: unreliable, so 96 stays
        self.overlap_wgrad = os.environ.get("DREAM_OVERLAP_WGRAD", "1") != "0"
        self.overlap_max_frames = int(os.environ.get("DREAM_OVERLAP_MAX_FRAMES", "96"))
        # the 3x3 convs' BatchNorm statistics / masked backward reductions in the Winograd F(2x2) kernel's epilogue (csrc/conv_wino.hip
        # WINO_STAT) -- 66 launches fewer per ResNet-101 step; measured round 5 (profiles/r05_ab_bn_fusion_3x3.txt, alternating on one
        # box): 350.4 -> 358.5 frames/s at 16 frames (+2.3 %), so on by default; "0" = the stand-alone bn_stats / bn_bwd_stats passes
        self.bn_fusion_3x3 = os.environ.get("DREAM_BN_FUSION_3X3", "1") == "1"
        # ... and (round 6) the last decoder BatchNorm's in the head conv's data gradient (run_backward_fused, "final")
        self.bn_fusion_head = os.environ.get("DREAM_BN_FUSION_HEAD", "1") == "1"
        self.stem_on_gemm = os.environ.get("DREAM_STEM_GEMM", "1") == "1"     # training: the 7x7 stem + its statistics + its weight gradient on the 1x1 GEMM
        self.ds_on_gemm = os.environ.get("DREAM_DS_GEMM", "1") == "1"         # training: the stride-2 downsample convs on the 1x1 GEMM over gathered pixels
        self.CONVT_GEMM_MAX_PIXELS = int(os.environ.get("DREAM_CONVT_GEMM_MAX_PIXELS", "6000"))   # training forward: transposed convs on small maps as GEMM + gather
        self.COL3_MAX_PIXELS = int(os.environ.get("DREAM_COL3_MAX_PIXELS", "12000"))   # ... and 3x3 stride-2 convs with at most this many output pixels
        # training: BatchNorm without its separate passes (round 4) -- statistics finished inside the launch that sums them (the
        # 1x1 convs' own epilogues where possible), BN + ReLU applied by the consuming 1x1 conv's loader, the backward reductions in
        # the data-gradient epilogue; "0" = the three-launch kernels of rounds 1-3 (A/B, tests)
        self.bn_fusion = os.environ.get("DREAM_BN_FUSION", "1") != "0"
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2)], 1):
            stage = nn.Sequential()
            for bi in range(blocks):
                blk = nn.Module()
                s = stride if bi == 0 else 1
                blk.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
                blk.bn1 = nn.BatchNorm2d(planes)
                blk.conv2 = nn.Conv2d(planes, planes, 3, stride=s, padding=1, bias=False)
                blk.bn2 = nn.BatchNorm2d(planes)
                blk.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
                blk.bn3 = nn.BatchNorm2d(planes * 4)
                if bi == 0 and (s != 1 or inplanes != planes * 4):
                    blk.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=s, bias=False),
                                                   nn.BatchNorm2d(planes * 4))
                inplanes = planes * 4
                stage.add_module(str(bi), blk)
            setattr(self, "layer%d" % li, stage)

        def up(ci):
            return [nn.ConvTranspose2d(ci, 256, 4, 2, 1, 0), nn.BatchNorm2d(256, momentum=0.1), nn.ReLU(inplace=True)]

        ups = up(2048) + up(256) + up(256) + up(256)
        if not full:
            self.upsample = nn.Sequential(*(ups + [nn.Conv2d(256, n_keypoints, 1, 1)]))
        else:
            self.upsample = nn.Sequential(*ups)
            self.upsample2 = nn.Sequential(*(up(256) + [nn.Conv2d(256, n_keypoints, 1, 1)]))
        # models.py:22: resnet101(pretrained=pretrained) -- ImageNet trunk when it can be had, one loud warning otherwise.
        # ``freeze`` is accepted and unused, exactly as in the reference (models.py:19: never read).
        self.imagenet_initialised = _pretrained.init_resnet101_trunk(self) if pretrained else False

    def output_resolution(self, input_wh):
        def trunk(v):
            v = (v + 2 * 3 - 7) // 2 + 1          # conv1 7x7 s2 p3
            v = (v + 2 * 1 - 3) // 2 + 1          # maxpool 3x3 s2 p1
            for _ in range(3):                    # layer2..4: 3x3 s2 p1
                v = (v + 2 * 1 - 3) // 2 + 1
            return v
        n_up = 5 if self.full else 4
        return tuple(trunk(int(v)) * (2 ** n_up) for v in input_wh)

    # ---- execution (eval-mode BatchNorm folded into the conv epilogues) ----------------------------------
    def _cached(self, key, tensors, build):
        rec = self.__dict__.get("_pack_records")
        if rec is not None and key[0] != "bn":
            return self._cached_recording(key, tensors, build, rec)
        pend = self.__dict__.get("_pack_pending")
        if pend is not None and key not in pend[1]:
            self._pack_join()                        # the late part of the step's re-pack runs on the second stream (_repack_weights)
        tag = tuple((t._version, t.data_ptr()) for t in tensors)
        hit = self._cache.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                hit = (tag, build())
            self._cache[key] = hit
        return hit[1]

    def _pack_join(self):
        """The main stream waits for the late part of this step's re-pack (second stream); a no-op when none is pending."""
        pend = self.__dict__.get("_pack_pending")
        if pend is not None:
            object.__setattr__(self, "_pack_pending", None)
            torch.cuda.current_stream().wait_event(pend[0])

    # ---- training: all packed weight copies of a step refreshed by ONE launch ------------------------------------------------------
    def _cached_recording(self, key, tensors, build, rec):
        """_cached during the recording step of _repack_weights: also notes which batchable packs build() made."""
        tag = tuple((t._version, t.data_ptr()) for t in tensors)
        hit = self._cache.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad(), ops.record_packs() as descs:
                out = build()
            self._cache[key] = hit = (tag, out)
            # batchable: everything build() launched was one of the table's kinds, reading the parameter itself (not a derived copy)
            if descs and all(any(d[1].data_ptr() == t.data_ptr() for t in tensors) for d in descs):
                rec[key] = (tensors, out, list(descs))
        return hit[1]

    def _repack_weights(self, split=False):
        """A training step re-packs every conv weight (the optimizer changed them all): 216 launches of 2-10 us, 6.8 % of a
        step at 16 frames (profiles/r02_layer_profile_resnet_h_train16.txt).  The first training step records which packed copies
        it builds from which parameter; from the second step on ONE launch (dream_pack_weights_batched: a device-resident job
        table, blockIdx.y = job) rewrites all of them in place at the start of the step and the cache entries are stamped with
        the parameters' new versions.  The few packs the table does not cover (tap-major copies for the direct kernel, the
        transposed convs' phase kernels) stay lazy.  Skipped while a hipGraph capture is under way (the data-parallel replicas
        capture their whole step, packing included)."""
        first = next(self.parameters())
        self._pack_join()
        if not first.is_cuda and not os.environ.get("DREAM_PACK_BATCHED_ON_CPU"):
            return
        if (first.is_cuda and torch.cuda.is_current_stream_capturing()) or os.environ.get("DREAM_PACK_BATCHED", "1") == "0":
            object.__setattr__(self, "_pack_records", None)
            return
        st = self.__dict__.get("_pack_state")
        stamp = first.data_ptr()
        if st is None or st["stamp"] != stamp:
            st = {"stamp": stamp, "table": None, "records": {}, "steps": 0}
            object.__setattr__(self, "_pack_state", st)
        if st["table"] is None:
            st["steps"] += 1
            if st["steps"] == 1 or not st["records"]:
                object.__setattr__(self, "_pack_records", st["records"])          # record this step's packs
                return
            object.__setattr__(self, "_pack_records", None)
            st["keys"] = list(st["records"].items())
            descs = [d for _, (_, _, ds) in st["keys"] for d in ds]
            st["table"], st["njobs"] = ops.pack_job_table(descs, first.device), len(descs)
            # the launch's workgroups by job size (round 6): the decoder's 2048 -> 256 transposed conv is 19 M packed floats, a 64 x 64 1x1 conv
            # 4 K -- with 16 workgroups each the large jobs set the launch's length (0.50 ms; DREAM_PACK_SPANS=0 restores that)
            st["spans"] = ops.pack_span_table(descs, first.device) if os.environ.get("DREAM_PACK_SPANS", "1") != "0" else None
            # Round 6, measured and NOT adopted (DREAM_PACK_SPLIT=1 opts in): the re-pack in two launches.  The copies the stem, layer1 and
            # layer2 read (3 % of the floats) are rewritten on the main stream; the rest -- layer3 on, the decoder, every data-gradient
            # operator -- on the SECOND stream, which is idle during a forward pass, while the main stream runs the first seven Bottlenecks
            # (~6 ms at 16 frames); the main stream waits for it at the first use of a late copy (_cached -> _pack_join).  One box,
            # alternating (profiles/r06_ab_pack_split.txt): 395.4 / 395.4 / 393.9 frames/s with one launch, 392.6 / 393.4 / 396.1 with two --
            # the 0.4-ms copy takes from the forward kernels what it saves (it moves 1.5 GB while they run).
            st["split"] = None
            if split and first.is_cuda and st["spans"] is not None and os.environ.get("DREAM_PACK_SPLIT", "0") == "1":
                early = [str(key[1]).startswith(("conv1", "layer1.", "layer2.")) for key, _ in st["keys"]]
                parts = []
                for want in (True, False):
                    ds = [d for (_, (_, _, dd)), e in zip(st["keys"], early) if e == want for d in dd]
                    parts.append((ops.pack_job_table(ds, first.device), ops.pack_span_table(ds, first.device)) if ds else None)
                if parts[0] is not None and parts[1] is not None:
                    st["split"] = (parts[0], parts[1], frozenset(key for (key, _), e in zip(st["keys"], early) if e))
        tags = {key: tuple((t._version, t.data_ptr()) for t in tensors) for key, (tensors, _, _) in st["keys"]}
        if any(self._cache.get(key, (None,))[0] != tag or self._cache[key][1] is not st["records"][key][1] for key, tag in tags.items()):
            if split and st.get("split") is not None:
                (t0, s0), (t1, s1), early_keys = st["split"]
                ops.pack_weights_spans(t0, s0[0], s0[1])
                main, side = torch.cuda.current_stream(), _SideStream.live_stream(first.device)
                side.wait_stream(main)                   # behind the optimizer step and the previous backward's reads of the old copies
                with torch.cuda.stream(side):
                    ops.pack_weights_spans(t1, s1[0], s1[1])
                    object.__setattr__(self, "_pack_pending", (side.record_event(), early_keys))
            elif st.get("spans") is not None:
                ops.pack_weights_spans(st["table"], st["spans"][0], st["spans"][1])
            else:
                ops.pack_weights_batched(st["table"], st["njobs"])
            for key, tag in tags.items():
                self._cache[key] = (tag, st["records"][key][1])

    def _fold(self, name, bn, conv_bias=None):
        tensors = [bn.weight, bn.bias, bn.running_mean, bn.running_var] + ([conv_bias] if conv_bias is not None else [])
        return self._cached(("bn", name), tensors, lambda: ops.bn_fold(
            bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps,
            conv_bias.detach() if conv_bias is not None else None))

    def _conv_bn(self, name, x, conv, bn, relu, residual=None):
        k, stride = int(conv.kernel_size[0]), int(conv.stride[0])
        scale, shift = self._fold(name, bn, conv.bias)
        cout, cin = int(conv.weight.shape[0]), int(conv.weight.shape[1])
        if k == 3 and stride == 1 and self.conv_algorithm == "winograd" and cin % 16 == 0 and cout >= 64:
            # the stride-1 3x3 convs of the bottlenecks: Winograd F(2x2,3x3) with the folded BatchNorm in the epilogue
            tile = ops.winograd_tile(int(x.shape[1]), int(x.shape[2]), cin, cout, int(x.shape[0]))
            u, rows = self._cached(("wino", name, tile), [conv.weight], lambda: ops.pack_weight_winograd_tile(conv.weight.detach(), 0, tile))
            return ops.conv3x3_winograd_tile(tile, x, u, rows, scale, shift, residual, CONV_RELU if relu else 0)
        if self._gemm1x1(conv, x):
            # the 1x1 convs of the bottlenecks: a plain GEMM without LDS (gemm1x1.hip), folded BatchNorm / residual / ReLU fused
            packed, rows = self._cached(("g0", name), [conv.weight], lambda: ops.pack_conv1x1_weight(conv.weight.detach(), 0))
            return ops.conv1x1(x, packed, rows, scale, shift, residual, CONV_RELU if relu else 0)
        if (self.ds_on_gemm and k == 1 and stride == 2 and self.conv1x1_algorithm == "gemm" and cin == int(x.shape[3]) and cin % 64 == 0):
            # the stride-2 downsample convs on the GEMM over the pixels they read (as in training: _unit_fused)
            xs = ops.subsample2(x)
            if ops.conv1x1_applies(xs, cout):
                packed, rows = self._cached(("g0", name), [conv.weight], lambda: ops.pack_conv1x1_weight(conv.weight.detach(), 0))
                return ops.conv1x1(xs, packed, rows, scale, shift, residual, CONV_RELU if relu else 0)
        packed, rows, _ = self._cached(("w", name), [conv.weight], lambda: ops.pack_conv_weight(conv.weight.detach(), 0))
        return ops.conv2d(x, packed, rows, k, stride, scale, shift, residual, CONV_RELU if relu else 0)

    def _convT_gemm(self, name, m, y):
        """Does this transposed conv run as one 1x1 GEMM (N = 16 Cout) + a gather (small maps: run_forward_train_fused)?  -> packed weight or None."""
        co_t, npx = int(m.weight.shape[1]), int(y.shape[0]) * int(y.shape[1]) * int(y.shape[2])
        if not (self.ds_on_gemm and self.conv1x1_algorithm == "gemm" and npx <= self.CONVT_GEMM_MAX_PIXELS and co_t % 4 == 0
                and tuple(m.kernel_size) == (4, 4) and tuple(m.stride) == (2, 2) and tuple(m.padding) == (1, 1)
                and tuple(m.output_padding) == (0, 0) and int(y.shape[3]) == int(m.weight.shape[0]) and ops.conv1x1_applies(y, 16 * co_t)):
            return None
        return self._cached(("g0T", name), [m.weight], lambda: ops.pack_conv1x1_weight(
            m.weight.detach().permute(2, 3, 1, 0).reshape(16 * co_t, -1, 1, 1).contiguous(), 0))

    def _gemm1x1(self, conv, x):
        """Stride-1 1x1 convs with channel counts the LDS-free GEMM kernel takes (all of ResNet-101's: multiples of 64)."""
        return (self.conv1x1_algorithm == "gemm" and int(conv.kernel_size[0]) == 1 and int(conv.stride[0]) == 1
                and int(conv.weight.shape[1]) == int(x.shape[3]) and ops.conv1x1_applies(x, int(conv.weight.shape[0])))

    # ---- inference on the split-precision conv kernel (strided convs stay on the fp32 kernel) ------------------
    def _conv_bn16(self, name, x, amax, conv, bn, relu, residual=None):
        """-> (y, amax_y).  Stride-1 1x1 / 3x3 convs run on conv_f16x3; the six stride-2 convs of the trunk run on
        the fp32 kernel, which publishes max|y| all the same."""
        k, stride = int(conv.kernel_size[0]), int(conv.stride[0])
        scale, shift = self._fold(name, bn, conv.bias)
        flags = CONV_RELU if relu else 0
        if stride != 1:
            packed, rows, _ = self._cached(("w", name), [conv.weight], lambda: ops.pack_conv_weight(conv.weight.detach(), 0))
            return ops.conv2d_amax(x, packed, rows, k, stride, scale, shift, residual, flags)
        p16 = self._cached(("w16", name), [conv.weight], lambda: ops.pack_conv_weight_f16x3(conv.weight.detach(), 0))
        return ops.conv2d_f16x3(x, amax, p16, p16[3], k, scale, shift, residual, flags)

    def run_forward_f16x3(self, x):
        col = ops.im2col_nchw(x, 7, 7, 2, 3, 160)
        amax = ops.absmax(x)                               # im2col only rearranges (and zero-pads) the image
        w1 = self._cached(("w16", "conv1"), [self.conv1.weight],
                          lambda: ops.pack_conv_weight_f16x3(self.conv1.weight.detach().reshape(64, 147, 1, 1), 0))
        s1, t1 = self._fold("bn1", self.bn1)
        y, amax = ops.conv2d_f16x3(col, amax, w1, 64, 1, s1, t1, None, CONV_RELU)
        del col
        y = ops.maxpool3s2(y)                              # pooling cannot raise the maximum: amax carries over
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self, "layer%d" % li)):
                name = "layer%d.%d" % (li, bi)
                idt = y
                if hasattr(blk, "downsample"):
                    idt, _ = self._conv_bn16(name + ".ds", y, amax, blk.downsample[0], blk.downsample[1], relu=False)
                o, a1 = self._conv_bn16(name + ".1", y, amax, blk.conv1, blk.bn1, relu=True)
                o, a2 = self._conv_bn16(name + ".2", o, a1, blk.conv2, blk.bn2, relu=True)
                y, amax = self._conv_bn16(name + ".3", o, a2, blk.conv3, blk.bn3, relu=True, residual=idt)
        seqs = [("upsample", self.upsample)] + ([("upsample2", self.upsample2)] if self.full else [])
        for sname, seq in seqs:
            mods = list(seq)
            i = 0
            while i < len(mods):
                m = mods[i]
                name = "%s.%d" % (sname, i)
                if isinstance(m, nn.ConvTranspose2d):
                    p16 = self._cached(("w16", name), [m.weight], lambda m=m: ops.pack_convT4x4_weight_f16x3(m.weight.detach()))
                    scale, shift = self._fold(name, mods[i + 1], m.bias)
                    y, amax = ops.conv_transpose4x4s2_f16x3(y, amax, p16, p16[3], scale, shift, CONV_RELU)
                    i += 3
                else:
                    p16 = self._cached(("w16", name), [m.weight], lambda m=m: ops.pack_conv_weight_f16x3(m.weight.detach(), 0))
                    y, _ = ops.conv2d_f16x3(y, amax, p16, p16[3], 1, None, m.bias.detach(), None, CONV_OUT_NCHW, want_amax=False)
                    i += 1
        return y

    def run_forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("expected [B,3,H,W] input, got %s" % (tuple(x.shape),))
        if self.precision == "fp16x3":
            return self.run_forward_f16x3(x)
        # stem: 7x7 s2 conv as im2col (K = 147 -> 160) + 1-tap MFMA conv, BN+ReLU fused; then MaxPool(3,2,1)
        col = ops.im2col_nchw(x, 7, 7, 2, 3, 160)
        s1, t1 = self._fold("bn1", self.bn1)
        if self.stem_on_gemm and self.conv1x1_algorithm == "gemm" and ops.conv1x1_applies(col, 64):
            # Round 6: the im2col rows (K = 160) through the 1x1 GEMM kernel (the 1-tap direct kernel ran them at 69 TFLOP/s; this is a
            # stream of 0.8 GB in, 0.3 GB out at 32 frames)
            def build():
                w2 = self.conv1.weight.detach().reshape(64, 147)
                return ops.pack_conv1x1_weight(torch.cat([w2, w2.new_zeros((64, 13))], dim=1).reshape(64, 160, 1, 1), 0)
            packed, rows = self._cached(("g0e", "conv1"), [self.conv1.weight], build)
            y = ops.conv1x1(col, packed, rows, s1, t1, None, CONV_RELU)
        else:
            w1 = self._cached(("w", "conv1"), [self.conv1.weight],
                              lambda: ops.pack_matrix_weight(self.conv1.weight.detach().reshape(64, 147), 160))
            y = ops.conv2d(col, w1[0], 64, 1, 1, s1, t1, None, CONV_RELU)
        del col
        y = ops.maxpool3s2(y)
        for li in (1, 2, 3, 4):
            stage = getattr(self, "layer%d" % li)
            for bi, blk in enumerate(stage):
                name = "layer%d.%d" % (li, bi)
                idt = y
                if hasattr(blk, "downsample"):
                    idt = self._conv_bn(name + ".ds", y, blk.downsample[0], blk.downsample[1], relu=False)
                o = self._conv_bn(name + ".1", y, blk.conv1, blk.bn1, relu=True)
                o = self._conv_bn(name + ".2", o, blk.conv2, blk.bn2, relu=True)
                y = self._conv_bn(name + ".3", o, blk.conv3, blk.bn3, relu=True, residual=idt)
        seqs = [("upsample", self.upsample)] + ([("upsample2", self.upsample2)] if self.full else [])
        for sname, seq in seqs:
            mods = list(seq)
            i = 0
            while i < len(mods):
                m = mods[i]
                name = "%s.%d" % (sname, i)
                if isinstance(m, nn.ConvTranspose2d):
                    bn = mods[i + 1]
                    scale, shift = self._fold(name, bn, m.bias)
                    gT = self._convT_gemm(name, m, y)
                    if gT is not None:
                        y = ops.col2im4s2(ops.conv1x1(y, gT[0], gT[1], None, None, None, ops.CONV_NO_KSPLIT), int(m.weight.shape[1]), shift, scale=scale, flags=CONV_RELU)
                    elif self.convT_algorithm == "winograd" and ops.convT4x4_winograd_applies(y, int(m.weight.shape[1])):
                        tile = ops.convT4x4_winograd_tile(y, int(m.weight.shape[1]))
                        u4, cout = self._cached(("wu4", name, tile), [m.weight], lambda m=m, tile=tile: ops.pack_convT4x4_winograd_weight_tile(m.weight.detach(), tile))
                        y = ops.conv_transpose4x4s2_winograd_tile(tile, y, u4, cout, scale, shift, CONV_RELU)
                    else:
                        packed, cout = self._cached(("w", name), [m.weight], lambda m=m: ops.pack_convT4x4_weight(m.weight.detach()))
                        y = ops.conv_transpose4x4s2(y, packed, cout, scale, shift, CONV_RELU)
                    i += 3                                  # ConvTranspose2d, BatchNorm2d, ReLU
                else:                                       # final 1x1 conv -> K belief maps, NCHW
                    packed, rows, _ = self._cached(("w", name), [m.weight], lambda m=m: ops.pack_conv_weight(m.weight.detach(), 0))
                    y = ops.conv2d(y, packed, rows, 1, 1, None, m.bias.detach(), None, CONV_OUT_NCHW)
                    i += 1
        return y

    # ---- training: train-mode BatchNorm (batch statistics), activations kept for the backward plan --------
    def _packed_w(self, name, conv, mode):
        return self._cached(("w%d" % mode, name), [conv.weight], lambda: ops.pack_conv_weight(conv.weight.detach(), mode))

    def _unit_fwd(self, tape, name, x, conv, bn, relu, residual=None):
        """conv -> BN(batch stats) (+residual) (ReLU); records what the backward needs."""
        k, stride = int(conv.kernel_size[0]), int(conv.stride[0])
        bias = conv.bias.detach() if conv.bias is not None else None
        if self._wino_train(conv):
            tile = ops.winograd_tile(int(x.shape[1]), int(x.shape[2]), int(conv.weight.shape[1]), int(conv.weight.shape[0]), int(x.shape[0]))
            u, rows = self._cached(("wino", name, tile), [conv.weight], lambda: ops.pack_weight_winograd_tile(conv.weight.detach(), 0, tile))
            z = ops.conv3x3_winograd_tile(tile, x, u, rows, None, bias, None, 0)
        elif self._gemm1x1(conv, x):
            packed, rows = self._cached(("g0", name), [conv.weight], lambda: ops.pack_conv1x1_weight(conv.weight.detach(), 0))
            z = ops.conv1x1(x, packed, rows, None, bias, None, 0)
        else:
            packed, rows, _ = self._packed_w(name, conv, 0)
            z = ops.conv2d(x, packed, rows, k, stride, None, bias, None, 0)
        y, mean, invstd = ops.bn_train_fwd(z, bn, residual, relu)
        tape.append(dict(kind="conv", name=name, conv=conv, bn=bn, relu=relu, x=x, z=z, y=y, mean=mean, invstd=invstd,
                         k=k, stride=stride, has_res=residual is not None))
        return y

    def _wino_train(self, conv):
        """Training: the stride-1 3x3 convs of the bottlenecks (forward and data gradient) on the Winograd kernel."""
        return (self.conv_algorithm == "winograd" and int(conv.kernel_size[0]) == 3 and int(conv.stride[0]) == 1
                and int(conv.weight.shape[1]) % 16 == 0 and int(conv.weight.shape[0]) % 16 == 0
                and min(int(conv.weight.shape[0]), int(conv.weight.shape[1])) >= 64)

    def _bwd_data(self, name, conv, dz, cin, k, stride, in_hw, residual=None):
        if self._wino_train(conv):
            tile = ops.winograd_tile(int(dz.shape[1]), int(dz.shape[2]), int(conv.weight.shape[0]), int(conv.weight.shape[1]), int(dz.shape[0]))
            u_t, rows = self._cached(("wino1", name, tile), [conv.weight], lambda: ops.pack_weight_winograd_tile(conv.weight.detach(), 1, tile))
            return ops.conv3x3_winograd_tile(tile, dz, u_t, rows, None, None, residual, 0)
        if (self.conv1x1_algorithm == "gemm" and k == 1 and stride == 1 and int(dz.shape[3]) == int(conv.weight.shape[0])
                and ops.conv1x1_applies(dz, cin)):
            packed_t, rows = self._cached(("g1", name), [conv.weight], lambda: ops.pack_conv1x1_weight(conv.weight.detach(), 1))
            return ops.conv1x1(dz, packed_t, rows, None, None, residual, 0)
        packed_t, rows, _ = self._packed_w(name, conv, 1)
        return ops.conv2d_bwd_data(dz, packed_t, cin, k, stride, in_hw, residual=residual)

    def run_forward_train(self, x):
        if self.bn_fusion:
            return self.run_forward_train_fused(x)
        self._repack_weights()
        tape = []
        col = ops.im2col_nchw(x, 7, 7, 2, 3, 160)
        w1 = self._cached(("w", "conv1"), [self.conv1.weight],
                          lambda: ops.pack_matrix_weight(self.conv1.weight.detach().reshape(64, 147), 160))
        z = ops.conv2d(col, w1[0], 64, 1, 1)
        y, mean, invstd = ops.bn_train_fwd(z, self.bn1, None, True)
        tape.append(dict(kind="stem", conv=self.conv1, bn=self.bn1, relu=True, x=col, z=z, y=y, mean=mean, invstd=invstd))
        yp = ops.maxpool3s2(y)
        tape.append(dict(kind="pool", x=y))
        y = yp
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self, "layer%d" % li)):
                name = "layer%d.%d" % (li, bi)
                tape.append(dict(kind="block_begin", name=name, ds=hasattr(blk, "downsample")))
                idt = y
                if hasattr(blk, "downsample"):
                    idt = self._unit_fwd(tape, name + ".ds", y, blk.downsample[0], blk.downsample[1], relu=False)
                o = self._unit_fwd(tape, name + ".1", y, blk.conv1, blk.bn1, relu=True)
                o = self._unit_fwd(tape, name + ".2", o, blk.conv2, blk.bn2, relu=True)
                y = self._unit_fwd(tape, name + ".3", o, blk.conv3, blk.bn3, relu=True, residual=idt)
                tape.append(dict(kind="block_end", name=name))
        seqs = [("upsample", self.upsample)] + ([("upsample2", self.upsample2)] if self.full else [])
        for sname, seq in seqs:
            mods = list(seq)
            i = 0
            while i < len(mods):
                m = mods[i]
                name = "%s.%d" % (sname, i)
                if isinstance(m, nn.ConvTranspose2d):
                    bn = mods[i + 1]
                    if self.convT_algorithm == "winograd" and ops.convT4x4_winograd_applies(y, int(m.weight.shape[1])):
                        tile = ops.convT4x4_winograd_tile(y, int(m.weight.shape[1]))
                        u4, cout = self._cached(("wu4", name, tile), [m.weight], lambda m=m, tile=tile: ops.pack_convT4x4_winograd_weight_tile(m.weight.detach(), tile))
                        z = ops.conv_transpose4x4s2_winograd_tile(tile, y, u4, cout, None, m.bias.detach(), 0)
                    else:
                        packed, cout = self._cached(("w", name), [m.weight], lambda m=m: ops.pack_convT4x4_weight(m.weight.detach()))
                        z = ops.conv_transpose4x4s2(y, packed, cout, None, m.bias.detach(), 0)
                    y2, mean, invstd = ops.bn_train_fwd(z, bn, None, True)
                    tape.append(dict(kind="convT", name=name, conv=m, bn=bn, relu=True, x=y, z=z, y=y2, mean=mean, invstd=invstd))
                    y = y2
                    i += 3
                else:
                    packed, rows, _ = self._packed_w(name, m, 0)
                    out = ops.conv2d(y, packed, rows, 1, 1, None, m.bias.detach(), None, CONV_OUT_NCHW)
                    tape.append(dict(kind="final", name=name, conv=m, x=y))
                    y = out
                    i += 1
        return y, tape

    def run_backward(self, tape, grad_out_nchw, reducer=None):
        """-> {parameter: gradient}.  Walks the tape backwards; gradients that meet at a Bottleneck input are summed
        by the residual input of the data-gradient conv (no separate add kernel)."""
        if tape and tape[0].get("fused"):
            return self.run_backward_fused(tape, grad_out_nchw, reducer)
        grads = _GradDict(reducer)
        # Measured (resnet_h, 400x400, one MI355X): +6.5 / +5.7 / +5.3 % at 16 / 32 / 64 frames, -2.7 % at 128, where
        # every kernel already fills the chip and the two streams only disturb each other's L2.
        stem_x = tape[0]["x"]                                  # im2col of the input: [B, H/2, W/2, 160]
        input_px = 4 * int(stem_x.shape[0]) * int(stem_x.shape[1]) * int(stem_x.shape[2])
        side = _SideStream.create(grad_out_nchw, self.overlap_wgrad and input_px <= self.overlap_max_frames * 400 * 400)
        g = None                 # gradient w.r.t. the output of the unit being processed
        block = None             # state of the Bottleneck being unwound
        for rec in reversed(tape):
            kind = rec["kind"]
            if kind == "final":
                m = rec["conv"]
                cout, cin = int(m.weight.shape[0]), int(m.weight.shape[1])
                gy = ops.nchw_to_nhwc(grad_out_nchw, cpad=ops.round_up(cout, 16))
                def leaf(m=m, x=rec["x"], gy=gy, cout=cout, cin=cin):
                    grads[m.weight], grads[m.bias] = ops.conv2d_wgrad(x, gy, cout, cin, 1, 1, 0, want_bias=True)
                _on_side(side, leaf, rec["x"], gy)
                packed_t, rows, _ = self._packed_w(rec["name"], m, 1)
                g = ops.conv2d(gy, packed_t, rows, 1, 1)
            elif kind == "convT":
                m, bn = rec["conv"], rec["bn"]
                dz, _, dgam, dbet = ops.bn_train_bwd(rec["z"], g, rec["y"], bn.weight, rec["mean"], rec["invstd"], True)
                grads[bn.weight], grads[bn.bias] = dgam, dbet
                def leaf(m=m, x=rec["x"], dz=dz):
                    if ops.convT4x4_wgrad_winograd_applies(x, dz):        # nine-position minimal filtering + the bias sums, one launch
                        grads[m.weight], grads[m.bias] = ops.convT4x4_wgrad_winograd(x, dz)
                    else:
                        grads[m.weight] = ops.convT4x4_wgrad(x, dz)
                        grads[m.bias] = ops.channel_sum(dz)
                _on_side(side, leaf, rec["x"], dz)
                cin_t, cout_t = int(m.weight.shape[0]), int(m.weight.shape[1])
                if (self.convT_algorithm == "winograd" and cout_t % 16 == 0 and cout_t >= 32 and cin_t > 64
                        and int(dz.shape[3]) == cout_t and dz.shape[1] % 2 == 0 and dz.shape[2] % 2 == 0):
                    tile = ops.conv4x4s2_winograd_tile_of(dz, cin_t)
                    u4b, rows = self._cached(("wu4b", rec["name"], tile), [m.weight], lambda m=m, tile=tile: ops.pack_convT4x4_winograd_weight_tile(m.weight.detach(), tile, 1))
                    g = ops.conv4x4s2_winograd_tile(tile, dz, u4b, rows)
                else:
                    pk, rows = self._cached(("wTb", rec["name"]), [m.weight], lambda m=m: ops.pack_convT4x4_bwd_weight(m.weight.detach()))
                    g = ops.conv4x4s2(dz, pk, rows)
            elif kind == "block_end":
                block = dict(g_out=g, g_idt=None, g_ds=None)
            elif kind == "conv":
                conv, bn, name = rec["conv"], rec["bn"], rec["name"]
                cout, cin = int(conv.weight.shape[0]), int(conv.weight.shape[1])
                is_ds = name.endswith(".ds")
                dy = block["g_idt"] if is_ds else g
                dz, gm, dgam, dbet = ops.bn_train_bwd(rec["z"], dy, rec["y"], bn.weight, rec["mean"], rec["invstd"],
                                                      rec["relu"], want_g=rec["has_res"])
                grads[bn.weight], grads[bn.bias] = dgam, dbet
                if rec["has_res"]:
                    block["g_idt"] = gm          # masked block-output gradient == gradient of the identity branch
                def leaf(conv=conv, x=rec["x"], dz=dz, cout=cout, cin=cin, k=rec["k"], stride=rec["stride"]):
                    if self._wino_train(conv) and ops.wgrad_winograd_pays(int(dz.shape[0]) * int(dz.shape[1]) * int(dz.shape[2]), cin, cout):
                        grads[conv.weight] = ops.conv3x3_wgrad_winograd(x, dz, cout, cin, want_bias=False)[0]
                    else:
                        if self.conv1x1_algorithm == "gemm" and k == 1 and stride == 1 and ops.conv1x1_wgrad_applies(x, dz, cout):
                            grads[conv.weight] = ops.conv1x1_wgrad(x, dz, cout, cin)
                        else:
                            grads[conv.weight] = ops.conv2d_wgrad(x, dz, cout, cin, k, stride)[0]
                _on_side(side, leaf, rec["x"], dz)
                in_hw = (int(rec["x"].shape[1]), int(rec["x"].shape[2]))
                if is_ds:
                    block["g_ds"] = self._bwd_data(name, conv, dz, cin, rec["k"], rec["stride"], in_hw)
                elif name.endswith(".1"):
                    # block input: main-path gradient + identity / downsample gradient
                    block["dz1"] = (name, conv, dz, cin, rec["k"], rec["stride"], in_hw)
                else:
                    g = self._bwd_data(name, conv, dz, cin, rec["k"], rec["stride"], in_hw)
            elif kind == "block_begin":
                name1, conv1, dz, cin, k, stride, in_hw = block["dz1"]
                other = block["g_ds"] if rec["ds"] else block["g_idt"]
                g = self._bwd_data(name1, conv1, dz, cin, k, stride, in_hw, residual=other)
                block = None
                _early_bucket_hook(reducer, rec["name"], side, lambda: [grads[p] for p in self._early_bucket_params()])
            elif kind == "pool":
                g = ops.maxpool3s2_bwd(g, rec["x"])
            elif kind == "stem":
                bn = rec["bn"]
                dz, _, dgam, dbet = ops.bn_train_bwd(rec["z"], g, rec["y"], bn.weight, rec["mean"], rec["invstd"], True)
                grads[bn.weight], grads[bn.bias] = dgam, dbet
                dw, _ = ops.conv2d_wgrad(rec["x"], dz, 64, 160, 1, 1)
                grads[rec["conv"].weight] = dw.reshape(64, 160)[:, :147].reshape(64, 3, 7, 7).contiguous()
                g = None
        if side is not None:
            side.join()
        return grads

    # ---- training, round 4: BatchNorm folded into its neighbours (csrc/bn.hip "round 4", csrc/gemm1x1.hip PRE / EPI) -------------
    def _ctr(self, device):
        """Allocator of zero ticket words for the BatchNorm launches of this replica (ops.bn_counter_buffer: every launch takes its own
        slice of one persistent buffer and leaves it zero; the cursor restarts with every forward pass)."""
        buf = self._cache.get(("bnctr",))
        if buf is None or buf.device != device:
            buf = self._cache[("bnctr",)] = ops.bn_counter_buffer(device)
            self._ctr_pos = 0

        def take(n):
            if n > buf.numel():                     # a short slice would let the kernel's ticket atomics run past the buffer
                raise RuntimeError("dream_amd: a BatchNorm launch needs %d ticket words, the replica's buffer holds %d" % (n, buf.numel()))
            if self._ctr_pos + n > buf.numel():
                self._ctr_pos = 0
            out = buf[self._ctr_pos:self._ctr_pos + n]
            self._ctr_pos += n
            return out
        return take

    def _unit_fused(self, tape, name, x, conv, bn, relu, residual=None, pre=None, materialize=True):
        """conv -> BatchNorm(batch statistics) (+ residual) (ReLU).  ``x``: the conv's input tensor, or -- with ``pre`` = the record of
        the producing unit -- that unit's un-normalised output z, whose BatchNorm + ReLU this conv applies while loading.
        ``materialize=False``: the normalised output is left to the consumer's loader (returns the record instead of a tensor)."""
        k, stride = int(conv.kernel_size[0]), int(conv.stride[0])
        bias = conv.bias.detach() if conv.bias is not None else None
        cout = int(conv.weight.shape[0])
        sub2 = None
        if (self.ds_on_gemm and k == 1 and stride == 2 and pre is None and self.conv1x1_algorithm == "gemm"
                and int(conv.weight.shape[1]) == int(x.shape[3]) and int(x.shape[3]) % 64 == 0):
            # Round 6: a stride-2 1x1 conv (the trunk's three downsample branches) reads every second pixel of every second row: gathered
            # once (ops.subsample2), it IS a stride-1 1x1 conv -- the GEMM kernel with the BatchNorm statistics in its epilogue instead of the
            # direct kernel + a statistics pass; its weight gradient and its data gradient (scattered back onto the block input's grid in
            # run_backward_fused) are GEMMs as well
            xs = ops.subsample2(x)
            if ops.conv1x1_applies(xs, cout):
                sub2, x, stride = (int(x.shape[1]), int(x.shape[2])), xs, 1
        col3 = None
        if (self.ds_on_gemm and k == 3 and stride == 2 and pre is None and self.conv1x1_algorithm == "gemm" and int(conv.padding[0]) == 1
                and int(conv.weight.shape[1]) == int(x.shape[3]) and int(x.shape[3]) % 64 == 0 and cout % 64 == 0
                and int(x.shape[0]) * ((int(x.shape[1]) - 1) // 2 + 1) * ((int(x.shape[2]) - 1) // 2 + 1) <= self.COL3_MAX_PIXELS):
            # ... and a 3x3 stride-2 conv with too few output pixels for the direct kernel to fill the chip (layer4.0.conv2 at 16 frames:
            # 2704; 36 TFLOP/s) runs on the GEMM over its patch rows (ops.im2col3s2: [pixels][9 Cin]), in all three directions as well
            col3 = (int(x.shape[1]), int(x.shape[2]))
            x, k, stride = ops.im2col3s2(x), 1, 1
        rec = dict(kind="conv", name=name, conv=conv, bn=bn, relu=relu, x=x, pre=pre, k=k, stride=stride, has_res=residual is not None,
                   y=None, sub2=sub2, col3=col3)
        if self._gemm1x1(conv, x) or sub2 is not None or col3 is not None:
            if col3 is not None:       # GEMM weight [Cout][t Cin + c] = w[co][c][ky][kx]
                packed, rows = self._cached(("g0", name), [conv.weight], lambda: ops.pack_conv1x1_weight(
                    conv.weight.detach().permute(0, 2, 3, 1).reshape(cout, -1, 1, 1).contiguous(), 0))
            else:
                packed, rows = self._cached(("g0", name), [conv.weight], lambda: ops.pack_conv1x1_weight(conv.weight.detach(), 0))
            rec["z"], rec["ab"], rec["mean"], rec["invstd"] = ops.conv1x1_bn(
                x, packed, rows, bn, self._ctr(x.device), pre_ab=None if pre is None else pre["ab"], shift=bias)
        else:
            assert pre is None
            z = None
            if self._wino_train(conv):
                tile = ops.winograd_tile(int(x.shape[1]), int(x.shape[2]), int(conv.weight.shape[1]), cout, int(x.shape[0]))
                u, rows = self._cached(("wino", name, tile), [conv.weight], lambda: ops.pack_weight_winograd_tile(conv.weight.detach(), 0, tile))
                if self.bn_fusion_3x3 and tile == 2 and cout % 64 == 0:
                    # the statistics ride in the F(2x2) kernel's epilogue: no separate pass over z
                    rec["z"], rec["ab"], rec["mean"], rec["invstd"] = ops.conv3x3_winograd_bn(x, u, rows, bn, self._ctr(x.device), shift=bias)
                else:
                    z = ops.conv3x3_winograd_tile(tile, x, u, rows, None, bias, None, 0)
            else:
                packed, rows, _ = self._packed_w(name, conv, 0)
                z = ops.conv2d(x, packed, rows, k, stride, None, bias, None, 0)
            if z is not None:
                rec["z"] = z
                rec["ab"], rec["mean"], rec["invstd"] = ops.bn_stats(z, bn, self._ctr(x.device))
        tape.append(rec)
        if not materialize:
            return rec
        rec["y"] = ops.bn_apply_ab(rec["z"], rec["ab"], residual, relu)
        return rec["y"]

    def run_forward_train_fused(self, x):
        self._repack_weights(split=True)
        self._ctr_pos = 0
        tape = [dict(kind="begin", fused=True)]
        ho, wo = (int(x.shape[2]) + 6 - 7) // 2 + 1, (int(x.shape[3]) + 6 - 7) // 2 + 1
        if self.stem_on_gemm and self.conv1x1_algorithm == "gemm" and (int(x.shape[0]) * ho * wo + 64) * 192 * 4 < (1 << 31):
            # Round 6: the 7x7 stem as a GEMM over im2col rows of 192 columns (147 taps, zero-padded to the 1x1 GEMM's granularity -- 160
            # for the direct kernel): the BatchNorm statistics ride in its epilogue (one pass over the 16 x 200 x 200 x 64 output less) and
            # its weight gradient runs on the GEMM-shaped kernel (0.56 -> ~0.2 ms on the MAIN stream: it is the last launch of a step)
            col = ops.im2col_nchw(x, 7, 7, 2, 3, 192)
            def build():
                w2 = self.conv1.weight.detach().reshape(64, 147)
                return ops.pack_conv1x1_weight(torch.cat([w2, w2.new_zeros((64, 45))], dim=1).reshape(64, 192, 1, 1), 0)
            packed, _ = self._cached(("g0", "conv1"), [self.conv1.weight], build)
            z, ab, mean, invstd = ops.conv1x1_bn(col, packed, 64, self.bn1, self._ctr(col.device))
        else:
            col = ops.im2col_nchw(x, 7, 7, 2, 3, 160)
            w1 = self._cached(("w", "conv1"), [self.conv1.weight],
                              lambda: ops.pack_matrix_weight(self.conv1.weight.detach().reshape(64, 147), 160))
            z = ops.conv2d(col, w1[0], 64, 1, 1)
            ab, mean, invstd = ops.bn_stats(z, self.bn1, self._ctr(z.device))
        y = ops.bn_apply_ab(z, ab, None, True)
        tape.append(dict(kind="stem", conv=self.conv1, bn=self.bn1, relu=True, x=col, z=z, y=y, ab=ab, mean=mean, invstd=invstd))
        yp, pidx = ops.maxpool3s2_idx(y)               # the backward pass compares one stored byte per window
        tape.append(dict(kind="pool", x=y, idx=pidx))
        y = yp
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self, "layer%d" % li)):
                name = "layer%d.%d" % (li, bi)
                tape.append(dict(kind="block_begin", name=name, ds=hasattr(blk, "downsample")))
                idt = y
                if hasattr(blk, "downsample"):
                    idt = self._unit_fused(tape, name + ".ds", y, blk.downsample[0], blk.downsample[1], relu=False)
                o = self._unit_fused(tape, name + ".1", y, blk.conv1, blk.bn1, relu=True)
                # conv2's BatchNorm + ReLU is applied by conv3's loader when conv3 runs on the GEMM kernel (always, for ResNet-101)
                r2 = self._unit_fused(tape, name + ".2", o, blk.conv2, blk.bn2, relu=True, materialize=False)
                # ... and when the BACKWARD GEMM applies too: conv3's data gradient (conv1x1_bwd_bnmask) reads dz3, which has 4x the
                # channels of z2 and crosses the kernel's 2-GB offset limit first; relu(BN(z2)) is never stored on this path, so the
                # decision has to be made here (round-4 advice: ~210 frames of 400x400 per GPU would pass the forward and raise in the backward)
                z2 = r2["z"]
                dz3_fits = int(z2.shape[0]) * int(z2.shape[1]) * int(z2.shape[2]) * int(blk.conv3.weight.shape[0]) * 4 < (1 << 31)
                if self._gemm1x1(blk.conv3, r2["z"]) and dz3_fits:
                    y = self._unit_fused(tape, name + ".3", r2["z"], blk.conv3, blk.bn3, relu=True, residual=idt, pre=r2)
                else:
                    r2["y"] = ops.bn_apply_ab(r2["z"], r2["ab"], None, True)
                    y = self._unit_fused(tape, name + ".3", r2["y"], blk.conv3, blk.bn3, relu=True, residual=idt)
                tape.append(dict(kind="block_end", name=name))
        seqs = [("upsample", self.upsample)] + ([("upsample2", self.upsample2)] if self.full else [])
        for sname, seq in seqs:
            mods = list(seq)
            i = 0
            while i < len(mods):
                m = mods[i]
                name = "%s.%d" % (sname, i)
                if isinstance(m, nn.ConvTranspose2d):
                    bn = mods[i + 1]
                    co_t = int(m.weight.shape[1])
                    gT = self._convT_gemm(name, m, y)
                    if gT is not None:
                        # Round 6: a transposed conv on a small map (the first decoder layer: 2048 -> 256 on 13 x 13 maps, 2704 pixels at 16
                        # frames -- 50-100 workgroups of the Winograd kernel on 256 CUs) as ONE 1x1 GEMM with N = 16 Cout (the sixteen tap
                        # contributions of every input pixel) + a gather that sums the <= 4 contributions landing on each output pixel
                        z = ops.col2im4s2(ops.conv1x1(y, gT[0], gT[1], None, None, None, ops.CONV_NO_KSPLIT), co_t, m.bias.detach() if m.bias is not None else None)
                    elif self.convT_algorithm == "winograd" and ops.convT4x4_winograd_applies(y, int(m.weight.shape[1])):
                        tile = ops.convT4x4_winograd_tile(y, int(m.weight.shape[1]))
                        u4, cout = self._cached(("wu4", name, tile), [m.weight], lambda m=m, tile=tile: ops.pack_convT4x4_winograd_weight_tile(m.weight.detach(), tile))
                        z = ops.conv_transpose4x4s2_winograd_tile(tile, y, u4, cout, None, m.bias.detach(), 0)
                    else:
                        packed, cout = self._cached(("w", name), [m.weight], lambda m=m: ops.pack_convT4x4_weight(m.weight.detach()))
                        z = ops.conv_transpose4x4s2(y, packed, cout, None, m.bias.detach(), 0)
                    ab, mean, invstd = ops.bn_stats(z, bn, self._ctr(z.device))
                    # Round 6: the LAST decoder layer's normalised activation is consumed by the head conv only -- which applies the
                    # BatchNorm + ReLU in its loader (and so do its weight gradient and the mask of its data gradient): never stored
                    head = mods[i + 3] if i + 3 < len(mods) else None
                    y2 = None if self._head_on_gemm(head, z) else ops.bn_apply_ab(z, ab, None, True)
                    tape.append(dict(kind="convT", name=name, conv=m, bn=bn, relu=True, x=y, z=z, y=y2, ab=ab, mean=mean, invstd=invstd))
                    y = y2
                    i += 3
                elif y is None:                                   # the head conv on the 1x1 GEMM, behind the BatchNorm loader
                    prev = tape[-1]
                    cout, cin = int(m.weight.shape[0]), int(m.weight.shape[1])
                    n4 = ops.round_up(cout, 4)
                    def build(m=m, n4=n4, cout=cout, cin=cin):
                        w2 = m.weight.detach().reshape(cout, cin)
                        return ops.pack_conv1x1_weight(torch.cat([w2, w2.new_zeros((n4 - cout, cin))]).reshape(n4, cin, 1, 1), 0)
                    packed, _ = self._cached(("g0h", name), [m.weight], build)
                    bias4 = self._cached(("g0hb", name), [m.bias], lambda m=m, n4=n4, cout=cout: torch.cat([m.bias.detach(), m.bias.new_zeros((n4 - cout,))]))
                    out = ops.nhwc_to_nchw(ops.conv1x1_pre(prev["z"], packed, n4, prev["ab"], bias4))
                    if n4 != cout:
                        out = out[:, :cout].contiguous()
                    tape.append(dict(kind="final", name=name, conv=m, x=None))
                    y = out
                    i += 1
                else:
                    packed, rows, _ = self._packed_w(name, m, 0)
                    out = ops.conv2d(y, packed, rows, 1, 1, None, m.bias.detach(), None, CONV_OUT_NCHW)
                    tape.append(dict(kind="final", name=name, conv=m, x=y))
                    y = out
                    i += 1
        return y, tape

    def _head_on_gemm(self, head, z):
        """Does the decoder's head conv (the module behind the last ConvTranspose2d + BatchNorm + ReLU) run on the 1x1 GEMM with that
        BatchNorm in its loader?  (a 1x1 conv from a multiple of 64 channels, tensors within the GEMM's 2-GB operand limit)"""
        return (self.bn_fusion_head and isinstance(head, nn.Conv2d) and self.conv1x1_algorithm == "gemm" and int(head.kernel_size[0]) == 1
                and int(head.stride[0]) == 1 and head.bias is not None and int(head.weight.shape[1]) == int(z.shape[3]) and int(z.shape[3]) % 64 == 0
                and (z.numel() // int(z.shape[3]) + 64) * int(z.shape[3]) * 4 < (1 << 31))

    def _bn_bwd_fused(self, rec, dy, want_g=False):
        """BatchNorm backward of a unit in two launches: (dgamma, dbeta) finished inside the reduction launch, then dz (and the masked
        gradient g when the Bottleneck's identity branch needs it).  The ReLU mask comes from the stored activation where the forward
        pass wrote one, else it is recomputed from (z, ab).  ``dy`` = ("masked", g, dgamma, dbeta): the consumer's data-gradient
        epilogue already masked and summed (conv1x1_bwd_bnmask) -- one launch."""
        bn = rec["bn"]
        dev, c = rec["z"].device, int(rec["z"].shape[3])
        if isinstance(dy, tuple):
            _, g, dgam, dbet = dy
            dz, _ = ops.bn_bwd_apply(rec["z"], g, bn.weight, rec["mean"], rec["invstd"], dgam, dbet)
            return dz, g, dgam, dbet
        y_act = rec["y"] if rec["relu"] else None
        ab = rec["ab"] if (rec["relu"] and y_act is None) else None
        dgam, dbet = ops.bn_bwd_stats(rec["z"], dy, rec["mean"], rec["invstd"], self._ctr(dev), y_act=y_act, ab=ab)
        dz, g = ops.bn_bwd_apply(rec["z"], dy, bn.weight, rec["mean"], rec["invstd"], dgam, dbet, y_act=y_act, ab=ab, want_g=want_g)
        return dz, g, dgam, dbet

    def run_backward_fused(self, tape, grad_out_nchw, reducer=None):
        """run_backward for a tape of run_forward_train_fused."""
        grads = _GradDict(reducer)
        stem_x = tape[1]["x"]
        input_px = 4 * int(stem_x.shape[0]) * int(stem_x.shape[1]) * int(stem_x.shape[2])
        side = _SideStream.create(grad_out_nchw, self.overlap_wgrad and input_px <= self.overlap_max_frames * 400 * 400)
        g = None
        block = None
        for idx in range(len(tape) - 1, -1, -1):
            rec = tape[idx]
            kind = rec["kind"]
            if kind == "final":
                m = rec["conv"]
                cout, cin = int(m.weight.shape[0]), int(m.weight.shape[1])
                gy = ops.nchw_to_nhwc(grad_out_nchw, cpad=ops.round_up(cout, 16))
                prev = tape[idx - 1] if idx >= 1 else None
                if rec["x"] is None:                              # the head ran behind the BatchNorm loader: so does its weight gradient
                    def leaf(m=m, prev=prev, gy=gy, cout=cout, cin=cin):
                        n4 = ops.round_up(cout, 4)
                        grads[m.weight] = ops.conv1x1_wgrad(prev["z"], gy, n4, cin, pre_ab=prev["ab"])[:cout]
                        grads[m.bias] = ops.channel_sum(gy)[:cout]
                    _on_side(side, leaf, prev["z"], gy, prev["ab"])
                else:
                    def leaf(m=m, x=rec["x"], gy=gy, cout=cout, cin=cin):
                        grads[m.weight], grads[m.bias] = ops.conv2d_wgrad(x, gy, cout, cin, 1, 1, 0, want_bias=True)
                    _on_side(side, leaf, rec["x"], gy)
                if (self.bn_fusion_head and prev is not None and prev["kind"] == "convT" and prev["y"] is rec["x"] and prev.get("ab") is not None
                        and self.conv1x1_algorithm == "gemm" and int(m.kernel_size[0]) == 1 and cin % 64 == 0
                        and tuple(prev["z"].shape) == tuple(gy.shape[:3]) + (cin,) and prev["z"].numel() < (1 << 31)
                        and (prev["z"].numel() // cin) * ops.round_up(cout, 32) * 4 < (1 << 31)):
                    # Round 6: the head's data gradient on the 1x1 GEMM with the LAST decoder BatchNorm's ReLU mask and its two backward sums
                    # in the epilogue (the trunk's form, conv1x1_bwd_bnmask): one launch instead of the direct conv + the stand-alone
                    # reduction pass over two 16 x 208 x 208 x 256 tensors.  The contraction is the K = 7 (17) keypoint channels, zero-padded
                    # to the GEMM's 32.
                    kp = ops.round_up(cout, 32)
                    gy32 = gy if int(gy.shape[3]) == kp else ops.nchw_to_nhwc(grad_out_nchw, cpad=kp)
                    def build(m=m, kp=kp, cout=cout, cin=cin):
                        w2 = m.weight.detach().reshape(cout, cin)
                        return ops.pack_conv1x1_weight(torch.cat([w2, w2.new_zeros((kp - cout, cin))]).reshape(kp, cin, 1, 1), 1)
                    packed_t, _ = self._cached(("g1h", rec["name"]), [m.weight], build)
                    gmask, dgh, dbh = ops.conv1x1_bwd_bnmask(gy32, packed_t, cin, prev["z"], prev["ab"], prev["mean"], prev["invstd"],
                                                             self._ctr(gy32.device), y_act=prev["y"])      # (y None: the mask from (z, ab))
                    g = ("masked", gmask, dgh, dbh)
                else:
                    packed_t, rows, _ = self._packed_w(rec["name"], m, 1)
                    g = ops.conv2d(gy, packed_t, rows, 1, 1)
            elif kind == "convT":
                m, bn = rec["conv"], rec["bn"]
                dz, _, dgam, dbet = self._bn_bwd_fused(rec, g)
                grads[bn.weight], grads[bn.bias] = dgam, dbet
                def leaf(m=m, x=rec["x"], dz=dz):
                    if ops.convT4x4_wgrad_winograd_applies(x, dz):        # nine-position minimal filtering + the bias sums, one launch
                        grads[m.weight], grads[m.bias] = ops.convT4x4_wgrad_winograd(x, dz)
                    else:
                        grads[m.weight] = ops.convT4x4_wgrad(x, dz)
                        grads[m.bias] = ops.channel_sum(dz)
                _on_side(side, leaf, rec["x"], dz)
                cin_t, cout_t = int(m.weight.shape[0]), int(m.weight.shape[1])
                if (self.convT_algorithm == "winograd" and cout_t % 16 == 0 and cout_t >= 32 and cin_t > 64
                        and int(dz.shape[3]) == cout_t and dz.shape[1] % 2 == 0 and dz.shape[2] % 2 == 0):
                    tile = ops.conv4x4s2_winograd_tile_of(dz, cin_t)
                    u4b, rows = self._cached(("wu4b", rec["name"], tile), [m.weight], lambda m=m, tile=tile: ops.pack_convT4x4_winograd_weight_tile(m.weight.detach(), tile, 1))
                    g = ops.conv4x4s2_winograd_tile(tile, dz, u4b, rows)
                else:
                    pk, rows = self._cached(("wTb", rec["name"]), [m.weight], lambda m=m: ops.pack_convT4x4_bwd_weight(m.weight.detach()))
                    g = ops.conv4x4s2(dz, pk, rows)
            elif kind == "block_end":
                block = dict(g_out=g, g_idt=None, g_ds=None)
            elif kind == "conv":
                conv, bn, name = rec["conv"], rec["bn"], rec["name"]
                cout, cin = int(conv.weight.shape[0]), int(conv.weight.shape[1])
                is_ds = name.endswith(".ds")
                dy = block["g_idt"] if is_ds else g
                dz, gm, dgam, dbet = self._bn_bwd_fused(rec, dy, want_g=rec["has_res"])
                grads[bn.weight], grads[bn.bias] = dgam, dbet
                if rec["has_res"]:
                    block["g_idt"] = gm          # masked block-output gradient == gradient of the identity branch
                pre = rec["pre"]
                def leaf(conv=conv, x=rec["x"], dz=dz, cout=cout, cin=cin, k=rec["k"], stride=rec["stride"], pre=pre, col3=rec.get("col3")):
                    if col3 is not None:         # the conv ran on its patch rows: dW [Cout][t Cin + c] -> [Cout][Cin][3][3]
                        dw2 = ops.conv1x1_wgrad(x, dz, cout, 9 * cin)
                        grads[conv.weight] = dw2.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
                    elif pre is not None:        # the conv's input was relu(BN(x)), applied by its loader: so does the weight gradient's
                        if not ops.conv1x1_wgrad_applies(x, dz, cout):
                            y_in = ops.bn_apply_ab(x, pre["ab"], None, True)
                            grads[conv.weight] = ops.conv2d_wgrad(y_in, dz, cout, cin, k, stride)[0]
                        else:
                            grads[conv.weight] = ops.conv1x1_wgrad(x, dz, cout, cin, pre_ab=pre["ab"])
                    elif self._wino_train(conv) and ops.wgrad_winograd_pays(int(dz.shape[0]) * int(dz.shape[1]) * int(dz.shape[2]), cin, cout):
                        grads[conv.weight] = ops.conv3x3_wgrad_winograd(x, dz, cout, cin, want_bias=False)[0]
                    elif self.conv1x1_algorithm == "gemm" and k == 1 and stride == 1 and ops.conv1x1_wgrad_applies(x, dz, cout):
                        grads[conv.weight] = ops.conv1x1_wgrad(x, dz, cout, cin)
                    else:
                        grads[conv.weight] = ops.conv2d_wgrad(x, dz, cout, cin, k, stride)[0]
                _on_side(side, leaf, *([rec["x"], dz] + ([pre["ab"]] if pre is not None else [])))
                in_hw = (int(rec["x"].shape[1]), int(rec["x"].shape[2]))
                if is_ds:
                    block["g_ds"] = self._bwd_data(name, conv, dz, cin, rec["k"], rec["stride"], in_hw)
                    if rec.get("sub2") is not None:       # the conv ran on the gathered pixels: its data gradient back on the input's grid
                        block["g_ds"] = ops.scatter2(block["g_ds"], *rec["sub2"])
                elif name.endswith(".1"):
                    block["dz1"] = (name, conv, dz, cin, rec["k"], rec["stride"], in_hw)
                elif pre is not None:
                    # data gradient + the ReLU mask and the two reductions of the producer's BatchNorm in ONE launch
                    packed_t, rows = self._cached(("g1", name), [conv.weight], lambda: ops.pack_conv1x1_weight(conv.weight.detach(), 1))
                    gmask, dg2, db2 = ops.conv1x1_bwd_bnmask(dz, packed_t, cin, pre["z"], pre["ab"], pre["mean"], pre["invstd"],
                                                            self._ctr(dz.device))
                    g = ("masked", gmask, dg2, db2)
                else:
                    prod = tape[idx - 1]
                    if (self.bn_fusion_3x3 and self._wino_train(conv) and cin % 64 == 0 and int(dz.shape[3]) == cout
                            and prod["kind"] == "conv" and prod["y"] is rec["x"] and prod["relu"] and not prod["has_res"]
                            and ops.winograd_tile(int(dz.shape[1]), int(dz.shape[2]), cout, cin, int(dz.shape[0])) == 2):
                        # data gradient + the ReLU mask and the two reductions of the producer's BatchNorm in ONE launch of the
                        # F(2x2) kernel (the mask is recomputed from the producer's (z, ab) exactly as its apply pass evaluated it)
                        u_t, _ = self._cached(("wino1", name, 2), [conv.weight], lambda: ops.pack_weight_winograd_tile(conv.weight.detach(), 1, 2))
                        gmask, dg1, db1 = ops.conv3x3_winograd_bwd_bnmask(dz, u_t, cin, prod["z"], prod["ab"], prod["mean"], prod["invstd"],
                                                                         self._ctr(dz.device))
                        g = ("masked", gmask, dg1, db1)
                    elif rec.get("col3") is not None:
                        # the conv ran on its patch rows: the GEMM's data gradient is the gradient of those rows, summed back onto the map
                        packed_t, rows = self._cached(("g1", name), [conv.weight], lambda: ops.pack_conv1x1_weight(
                            conv.weight.detach().permute(0, 2, 3, 1).reshape(cout, -1, 1, 1).contiguous(), 1))
                        g = ops.col2im3s2(ops.conv1x1(dz, packed_t, rows, None, None, None, 0), *rec["col3"])
                    else:
                        g = self._bwd_data(name, conv, dz, cin, rec["k"], rec["stride"], in_hw)
            elif kind == "block_begin":
                name1, conv1, dz, cin, k, stride, in_hw = block["dz1"]
                other = block["g_ds"] if rec["ds"] else block["g_idt"]
                # the block's input is the previous Bottleneck's output relu(BN3(z3) + identity): that BatchNorm's ReLU mask and its
                # two backward reductions ride in the epilogue of this data-gradient GEMM (one full pass over three 4x-wide tensors
                # and one launch fewer per Bottleneck)
                prev = tape[idx - 2] if idx >= 2 and tape[idx - 1]["kind"] == "block_end" else None
                if (prev is not None and prev["kind"] == "conv" and prev["has_res"] and prev["relu"] and prev["y"] is not None
                        and self.conv1x1_algorithm == "gemm" and k == 1 and stride == 1 and int(dz.shape[3]) == int(conv1.weight.shape[0])
                        and ops.conv1x1_applies(dz, cin) and tuple(prev["y"].shape) == tuple(dz.shape[:3]) + (cin,)):
                    packed_t, rows = self._cached(("g1", name1), [conv1.weight], lambda: ops.pack_conv1x1_weight(conv1.weight.detach(), 1))
                    gmask, dg3, db3 = ops.conv1x1_bwd_bnmask(dz, packed_t, cin, prev["z"], None, prev["mean"], prev["invstd"],
                                                            self._ctr(dz.device), y_act=prev["y"], residual=other)
                    g = ("masked", gmask, dg3, db3)
                else:
                    g = self._bwd_data(name1, conv1, dz, cin, k, stride, in_hw, residual=other)
                block = None
                _early_bucket_hook(reducer, rec["name"], side, lambda: [grads[p] for p in self._early_bucket_params()])
            elif kind == "pool":
                g = ops.maxpool3s2_idx_bwd(g, rec["idx"], rec["x"].shape)
            elif kind == "stem":
                bn = rec["bn"]
                dz, _, dgam, dbet = self._bn_bwd_fused(rec, g)
                grads[bn.weight], grads[bn.bias] = dgam, dbet
                kcol = int(rec["x"].shape[3])                   # 192: the stem ran on the 1x1 GEMM (run_forward_train_fused), 160: direct
                if kcol == 192:
                    dw = ops.conv1x1_wgrad(rec["x"], dz, 64, 192)
                else:
                    dw, _ = ops.conv2d_wgrad(rec["x"], dz, 64, 160, 1, 1)
                grads[rec["conv"].weight] = dw.reshape(64, kcol)[:, :147].reshape(64, 3, 7, 7).contiguous()
                g = None
        if side is not None:
            side.join()
        return grads

    def dp_parameters(self):
        return list(self.parameters())

    def dp_trainable(self):
        return self.training                     # evaluation mode has no backward plan (BatchNorm folded)

    def dp_forward(self, x, save):
        if self.training:
            out, tape = self.run_forward_train(x)
            return [out], (tape if save else None)
        return [self.run_forward(x)], None

    EARLY_BUCKET_FROM = "layer3"                 # the early bucket of the single-process exchange: parameters from this layer on

    def dp_early_bucket(self):
        """-> (index k into dp_parameters() of the first parameter of the early bucket, marker of the backward plan) or None: the
        gradients of dp_parameters()[k:] are final once the backward plan has passed the block ``marker`` (the first Bottleneck of
        EARLY_BUCKET_FROM; its own BatchNorm gradients were produced by the blocks behind it)."""
        names = [n for n, _ in self.named_parameters()]
        ks = [i for i, n in enumerate(names) if n.startswith(self.EARLY_BUCKET_FROM + ".")]
        if not ks or ks[0] == 0:
            return None
        return ks[0], self.EARLY_BUCKET_FROM + ".0"

    def _early_bucket_params(self):
        k = self.dp_early_bucket()[0]
        return list(self.parameters())[k:]

    def dp_backward(self, tape, grad_outs, reducer=None):
        gdict = self.run_backward(tape, grad_outs[0].contiguous(), reducer=reducer)
        return [gdict[p] for p in self.parameters()]

    def dp_finish(self, outs):
        return outs

    def forward(self, x):
        params = list(self.parameters())
        if self.training:
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                return [_ResnetFunction.apply(self, x, *params)]
            with torch.no_grad():
                return [self.run_forward_train(x)[0]]
        with torch.no_grad():
            return [self.run_forward(x)]


class _ResnetFunction(torch.autograd.Function):
    """Whole-network autograd node for ResnetSimple in training mode (see _HourglassFunction)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        out, tape = module.run_forward_train(x.detach())
        ctx.module, ctx.tape, ctx.params = module, tape, params
        return out

    @staticmethod
    def backward(ctx, grad_out):
        reducer = _OverlappedAllReduce.create()
        gdict = _guarded_backward(ctx.module.run_backward, ctx.tape, grad_out.contiguous(), reducer=reducer)
        ctx.tape = None
        return (None, None) + tuple(_reduced(reducer, [gdict[p] for p in ctx.params]))


from .data_parallel import DreamDataParallel  # noqa: E402,F401  (the reference's torch.nn.DataParallel wrapper, network.py:244-256)
