"""torch-CPU restatement of the reference's belief-map CNNs.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Used as the parity checker for the HIP
path and as the ``cpu_baseline`` leg of bench.py; never imported by ``dream_amd``.

Restates (state_dict keys and shapes identical, so weights interchange):
  * DreamHourglass   -- /root/reference/dream/models.py:557-827  (vgg_q / vgg_f and flags)
  * ResnetSimple     -- /root/reference/dream/models.py:17-155   (resnet_h / resnet_f)
  * SoftArgmaxPavlo  -- /root/reference/dream/spatial_softmax.py:15-95
Pinned by tests/golden/cnn_*.npz (stub-imported reference, same weights, same inputs).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .topology import vgg19_features, ResNet101


def _named_seq(items):
    seq = nn.Sequential()
    for name, mod in items:
        seq.add_module(str(name), mod)
    return seq


class SoftArgmaxPavlo(nn.Module):
    """spatial_softmax.py:15-95: 7x7 avg-pool (zero pad, /49) -> per-map max-subtract ->
    exp(beta*.) -> / (sum + 1e-8) -> expectation of column / row index."""

    def __init__(self, n_keypoints=5, learned_beta=False, initial_beta=25.0):
        super().__init__()
        beta = torch.ones(n_keypoints) * initial_beta
        if learned_beta:
            self.beta = nn.Parameter(beta)          # spatial_softmax.py:19-20
        else:
            self.beta = beta                        # plain tensor, not in state_dict (:22)

    def forward(self, heatmaps, size_mult=1.0):
        b, k, h, w = heatmaps.shape
        pooled = F.avg_pool2d(heatmaps, 7, stride=1, padding=3)          # :18,:35
        flat = pooled.contiguous().view(b, k, -1)
        flat = flat - flat.max(dim=2, keepdim=True)[0]                   # :41-47
        e = torch.exp(self.beta.view(1, k, 1) * flat)                    # :49-52
        p = e.view(b, k, h, w) / (e.sum(dim=2, keepdim=True).view(b, k, 1, 1) + 1e-8)  # :55-59
        cols = (torch.arange(0, w) * size_mult).float().view(1, 1, 1, w)
        rows = (torch.arange(0, h) * size_mult).float().view(1, 1, h, 1)
        x = (p * cols).view(b, k, -1).sum(dim=2)                         # :82-85
        y = (p * rows).view(b, k, -1).sum(dim=2)                         # :87-90
        return torch.stack((x, y), dim=2)                                # :92


class DreamHourglass(nn.Module):
    """models.py:557-827.  VGG19 encoder blocks (torchvision indices 0-3, 5-8, 10-17, 19-26,
    28-35 with a fresh first conv), own MaxPool2d(2) between them, then either the
    nearest-upsample decoder (Q) or the ConvTranspose decoder (F), then the 3-conv head."""

    def __init__(self, n_keypoints, n_image_input_channels=3, internalize_spatial_softmax=True,
                 learned_beta=True, initial_beta=1.0, skip_connections=False,
                 deconv_decoder=False, full_output=False):
        super().__init__()
        self.n_keypoints = n_keypoints
        self.internalize_spatial_softmax = internalize_spatial_softmax
        self.skip_connections = skip_connections
        self.deconv_decoder = deconv_decoder
        self.full_output = full_output
        vgg = vgg19_features()
        self.down_sample = nn.MaxPool2d(2)                                             # :589

        first = [(0, nn.Conv2d(n_image_input_channels, 64, 3, 1, 1))]                  # :592-597
        self.layer_0_1_down = _named_seq(first + [(i, vgg[i]) for i in range(1, 4)])   # :598-599
        self.layer_0_2_down = _named_seq([(i, vgg[i]) for i in range(5, 9)])           # :601-603
        self.layer_0_3_down = _named_seq([(i, vgg[i]) for i in range(10, 18)])         # :605-607
        self.layer_0_4_down = _named_seq([(i, vgg[i]) for i in range(19, 27)])         # :609-611
        self.layer_0_5_down = _named_seq([(i, vgg[i]) for i in range(28, 36)])         # :613-615

        def conv(ci, co):
            return nn.Conv2d(ci, co, 3, 1, 1)

        def deconv(ci, co):
            return nn.ConvTranspose2d(ci, co, (3, 3), (2, 2), padding=1, output_padding=1)

        relu = lambda: nn.ReLU(inplace=True)
        if deconv_decoder:                                                             # :618-686
            self.deconv_0_4 = _named_seq([(0, deconv(512, 256)), (1, relu()), (2, conv(256, 256)), (3, relu())])
            self.deconv_0_3 = _named_seq([(0, deconv(256, 128)), (1, relu()), (2, conv(128, 128)), (3, relu())])
            self.deconv_0_2 = _named_seq([(0, deconv(128, 64)), (1, relu()), (2, conv(64, 64)), (3, relu())])
            self.deconv_0_1 = _named_seq([(0, deconv(64, 64)), (1, relu())])
        else:                                                                          # :688-733
            up = lambda: nn.Upsample(scale_factor=2)
            # note: both blocks end WITHOUT a ReLU (:698-700, :708-710)
            self.upsample_0_4 = _named_seq([(0, up()), (4, conv(512, 256)), (5, relu()), (6, conv(256, 256))])
            self.upsample_0_3 = _named_seq([(0, up()), (4, conv(256, 128)), (5, relu()), (6, conv(128, 64))])
            if full_output:
                self.upsample_0_2 = _named_seq([(0, up()), (2, conv(64, 64)), (3, relu()), (4, conv(64, 64)), (5, relu())])
                self.upsample_0_1 = _named_seq([("00", up()), (2, conv(64, 64)), (3, relu()), (4, conv(64, 64)), (5, relu())])
        self.heads_0 = _named_seq([(0, conv(64, 64)), (1, relu()), (2, conv(64, 32)), (3, relu()),
                                   (4, conv(32, n_keypoints))])                        # :736-747
        if internalize_spatial_softmax:                                                # :750-759
            self.softmax = _named_seq([(0, SoftArgmaxPavlo(n_keypoints, learned_beta, initial_beta))])

    def forward(self, x):                                                              # :761-827
        x1 = self.layer_0_1_down(x)
        x1d = self.down_sample(x1)
        x2 = self.layer_0_2_down(x1d)
        x2d = self.down_sample(x2)
        x3 = self.layer_0_3_down(x2d)
        x3d = self.down_sample(x3)
        x4 = self.layer_0_4_down(x3d)
        x4d = self.down_sample(x4)
        x5 = self.layer_0_5_down(x4d)
        skip = self.skip_connections
        d = x5 + x4d if skip else x5
        if self.deconv_decoder:
            y = self.deconv_0_4(d)
            y = self.deconv_0_3(y + x3d if skip else y)
            y = self.deconv_0_2(y + x2d if skip else y)
            y = self.deconv_0_1(y + x1d if skip else y)
            out = self.heads_0(y + x1 if skip else y)
        else:
            y = self.upsample_0_4(d)
            y = self.upsample_0_3(y + x3d if skip else y)
            if self.full_output:
                y = self.upsample_0_1(self.upsample_0_2(y))
            out = self.heads_0(y)
        outs = [out]
        if self.internalize_spatial_softmax:
            outs.append(self.softmax(out))
        return outs


class DreamHourglassMultiStage(nn.Module):
    """models.py:350-553.  S in [1,6] hourglasses; stage s>1 sees cat([image, maps of stage s-1]) with the maps
    nearest-upsampled x4 unless the decoder already reaches input resolution (:487-493); returns the S belief-map
    tensors (the soft-argmax head of each stage is computed and dropped, :481)."""

    def __init__(self, n_keypoints, n_image_input_channels=3, internalize_spatial_softmax=True, learned_beta=True,
                 initial_beta=1.0, n_stages=2, skip_connections=False, deconv_decoder=False, full_output=False):
        super().__init__()
        assert isinstance(n_stages, int), \
            'Expected "n_stages" to be an integer, but it is {}.'.format(type(n_stages))
        assert 0 < n_stages and n_stages <= 6, \
            "DreamHourglassMultiStage can only be constructed with 1 to 6 stages at this time."
        self.num_stages = n_stages
        self.deconv_decoder = deconv_decoder
        self.full_output = full_output
        for s in range(1, n_stages + 1):
            cin = n_image_input_channels + (n_keypoints if s > 1 else 0)                       # :400-478
            setattr(self, "stage%d" % s, DreamHourglass(
                n_keypoints, cin, internalize_spatial_softmax, learned_beta, initial_beta,
                skip_connections=skip_connections, deconv_decoder=deconv_decoder, full_output=full_output))

    def forward(self, x):                                                                      # :480-553
        outs = [self.stage1(x)[0]]
        for s in range(2, self.num_stages + 1):
            prev = outs[-1]
            if not (self.deconv_decoder or self.full_output):
                prev = F.interpolate(prev, scale_factor=4)
            outs.append(getattr(self, "stage%d" % s)(torch.cat([x, prev], dim=1))[0])
        return outs


class ResnetSimple(nn.Module):
    """models.py:17-155.  ResNet101 trunk, 4 (or 5) x [ConvT 4x4 s2 p1 -> BN -> ReLU], 1x1 -> K."""

    def __init__(self, n_keypoints=7, full=False):
        super().__init__()
        net = ResNet101()
        self.full = full
        self.conv1, self.bn1, self.relu, self.maxpool = net.conv1, net.bn1, net.relu, net.maxpool
        self.layer1, self.layer2, self.layer3, self.layer4 = net.layer1, net.layer2, net.layer3, net.layer4

        def up_block(ci):
            return [nn.ConvTranspose2d(ci, 256, 4, 2, 1, 0), nn.BatchNorm2d(256, momentum=0.1),
                    nn.ReLU(inplace=True)]

        ups = up_block(2048) + up_block(256) + up_block(256) + up_block(256)
        if not full:
            self.upsample = nn.Sequential(*(ups + [nn.Conv2d(256, n_keypoints, 1, 1)]))       # :37-79
        else:
            self.upsample = nn.Sequential(*ups)                                               # :81-122
            self.upsample2 = nn.Sequential(*(up_block(256) + [nn.Conv2d(256, n_keypoints, 1, 1)]))  # :124-136

    def forward(self, x):                                                                     # :138-155
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.upsample(x)
        if self.full:
            x = self.upsample2(x)
        return [x]


# Architecture-block overrides (on the shipped vgg_q / vgg_f YAML) that reach the remaining constructor branches of
# network.py:194-256: skip connections, the full-resolution upsample decoder, the soft-argmax head, multi-stage.
VARIANTS = {
    "vgg_q_skip": ("vgg_q", {"skip_connections": True}),
    "vgg_f_skip": ("vgg_f", {"skip_connections": True}),
    "vgg_full": ("vgg_q", {"deconv_decoder": False, "full_output": True}),
    "vgg_q_softmax": ("vgg_q", {"spatial_softmax": {"learned_beta": True, "initial_beta": 25.0},
                                "output_heads": ["belief_maps", "keypoints"]}),
    "vgg_ms2": ("vgg_q", {"n_stages": 2}),                      # n_stages only read with full_output (:225-230) -> default 2
    "vgg_f_ms2_skip": ("vgg_f", {"n_stages": 2, "skip_connections": True}),
    "vgg_ms3_full": ("vgg_q", {"deconv_decoder": False, "full_output": True, "n_stages": 3}),
}


def build_variant(name, n_keypoints):
    """What DreamNetwork.__init__ (network.py:194-256) builds for VARIANTS[name], minus DataParallel."""
    base, over = VARIANTS[name]
    kw = {"internalize_spatial_softmax": False}
    if "spatial_softmax" in over:
        kw = {"internalize_spatial_softmax": True, "learned_beta": over["spatial_softmax"]["learned_beta"],
              "initial_beta": over["spatial_softmax"]["initial_beta"]}
    deconv = over.get("deconv_decoder", base == "vgg_f")
    kw["deconv_decoder"] = deconv
    if "full_output" in over:
        kw["full_output"] = True
        if "n_stages" in over:
            kw["n_stages"] = over["n_stages"]
    if "skip_connections" in over:
        kw["skip_connections"] = over["skip_connections"]
    cls = DreamHourglassMultiStage if "n_stages" in over else DreamHourglass
    return cls(n_keypoints, **kw)


def build_model(arch, n_keypoints):
    """arch in {vgg_q, vgg_f, resnet_h, resnet_f}: what DreamNetwork.__init__ builds for the four
    shipped arch_configs (network.py:194-284), minus the DataParallel wrapper."""
    if arch in VARIANTS:
        return build_variant(arch, n_keypoints)
    if arch == "vgg_q":
        return DreamHourglass(n_keypoints, internalize_spatial_softmax=False)
    if arch == "vgg_f":
        return DreamHourglass(n_keypoints, internalize_spatial_softmax=False, deconv_decoder=True)
    if arch == "resnet_h":
        return ResnetSimple(n_keypoints, full=False)
    if arch == "resnet_f":
        return ResnetSimple(n_keypoints, full=True)
    raise ValueError(arch)


def recipe_weights(state_dict, final_keys=(), final_scale=1.0):
    """Construction-order-independent synthetic weights: every tensor is
    drawn from RandomState(crc32(key)), fan-in scaled so activations neither die nor blow up.
    Returns a new dict; BN running stats are made non-trivial so eval-mode BN is exercised."""
    import zlib
    import numpy as np
    out = {}
    for key, t in state_dict.items():
        rs = np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)
        shape = tuple(t.shape)
        if key.endswith("num_batches_tracked"):
            v = np.zeros(shape, dtype=np.int64)
        elif key.endswith("running_mean"):
            v = rs.uniform(-0.1, 0.1, shape)
        elif key.endswith("running_var"):
            v = rs.uniform(0.8, 1.2, shape)
        elif t.dim() == 4:
            # He-uniform on fan-in keeps ReLU activations O(1) through 20+ layers
            if "deconv" in key or ("upsample" in key and t.shape[-1] == 4):
                # ConvTranspose2d weight is [Cin, Cout, kh, kw]; each output sees ~kh*kw/4 taps
                fan_in = t.shape[0] * t.shape[2] * t.shape[3] / 4.0
            else:
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            bound = (6.0 / fan_in) ** 0.5
            v = rs.uniform(-bound, bound, shape)
        elif key.endswith("weight"):        # BN gamma (last BN of a bottleneck kept small so the
            v = rs.uniform(0.8, 1.2, shape)  # 33 residual adds do not blow the activations up)
            if "bn3." in key or "downsample.1." in key:
                v = v * 0.25
        elif key.endswith("beta"):
            v = np.asarray(t.detach().cpu().numpy(), dtype=np.float64)
        else:                               # biases / BN beta
            v = rs.uniform(-0.05, 0.05, shape)
        if any(key.endswith(fk) for fk in final_keys):
            v = v * final_scale
        out[key] = torch.as_tensor(np.asarray(v), dtype=t.dtype)
    return out


def smooth_weights(state_dict):
    """Weights for the round-5 structured VGG fixtures (tests/golden/structured_vgg_q_400.npz, structured_vgg_f.npz): every conv is a
    three-input NON-NEGATIVE channel mix (rows summing to one) times the binomial kernel [1,2,1] x [1,2,1] / 16, without bias -- a smoothing,
    mean-preserving layer; the 3-channel first conv carries signed colour filters with a small positive bias, so its ReLU outputs are
    colour-selective; stride-2 transposed convs (vgg_f) carry the bilinear kernel [1,2,1] x [1,2,1] / 4.  A blob in the frame then comes
    out as ONE smooth bump in every belief map, with a height that depends on its colour and the keypoint's channel mix: most maps
    have a clear winner (a detection), two comparable blobs give a rejection by the 0.25 rule.  (The recipe weights give ripples
    around every blob -- several local maxima within 0.25 of each other -- and 2 detections out of 14 at 400 x 400.)  The last
    layer is calibrated and stored by make_golden.py."""
    import zlib
    import numpy as np
    out = {}
    binom = np.outer([1.0, 2.0, 1.0], [1.0, 2.0, 1.0])
    for key, t in state_dict.items():
        rs = np.random.RandomState(zlib.crc32(("smooth:" + key).encode()) & 0x7FFFFFFF)
        shape = tuple(t.shape)
        if t.dim() == 4:
            transposed = "deconv" in key
            ci = shape[0] if transposed else shape[1]
            co = shape[1] if transposed else shape[0]
            if ci <= 4:                                     # first conv: signed colour filters of four TYPES (channel o: type o % 4)
                types = rs.uniform(-1.0, 1.0, (4, ci))
                mix = types[np.arange(co) % 4] * rs.uniform(0.8, 1.2, (co, 1))
            else:
                # output o mixes inputs o, o + 4, o + 8 (mod ci): channels of one colour type only, so the colour preference of the
                # first layer survives the depth
                mix = np.zeros((co, ci))
                for j, wgt in enumerate((0.6, 0.25, 0.15)):
                    mix[np.arange(co), (np.arange(co) + 4 * j) % ci] += wgt * rs.uniform(0.8, 1.2, co)
                mix = mix / mix.sum(1, keepdims=True)
            assert shape[2:] == (3, 3), (key, shape)
            kern = binom / (4.0 if transposed else 16.0)
            v = mix[:, :, None, None] * kern[None, None]
            if transposed:
                v = v.transpose(1, 0, 2, 3)
        elif key.endswith(".bias") and any(k2 == key[:-len("bias")] + "weight" and tuple(t2.shape)[1] <= 4 and t2.dim() == 4
                                           for k2, t2 in state_dict.items()):
            v = rs.uniform(0.05, 0.3, shape)                 # first conv
        else:
            v = np.zeros(shape)
        out[key] = torch.as_tensor(np.ascontiguousarray(v), dtype=t.dtype)
    return out


def structured_weights(state_dict):
    """Variant of recipe_weights for the structured end-to-end fixture of ResnetSimple (tests/golden/structured_resnet_h.npz):
    recipe weights, except that (i) every additive term is removed (conv / BN biases and BN running means are zero), so a
    zero background -- zero padding included -- maps to exactly zero and the belief maps are driven by the image content
    alone, and (ii) the 4x4 stride-2 transposed convs of the decoder carry a random channel mix times the bilinear
    kernel [1,3,3,1] x [1,3,3,1] / 16 (a partition of unity over the output phases: no checkerboard), so a blob in the
    image comes out as a blob in the maps.  The last (1x1) layer is calibrated and stored by make_golden.py."""
    import zlib
    import numpy as np
    out = recipe_weights(state_dict)
    bil = np.outer([1.0, 3.0, 3.0, 1.0], [1.0, 3.0, 3.0, 1.0]) / 16.0
    for key, t in out.items():
        if t.dim() == 4 and t.shape[-1] == 4 and "upsample" in key:
            ci, co = int(t.shape[0]), int(t.shape[1])
            rs = np.random.RandomState(zlib.crc32(("mix:" + key).encode()) & 0x7FFFFFFF)
            mix = rs.uniform(-1.0, 1.0, (ci, co)) * (6.0 / ci) ** 0.5
            out[key] = torch.as_tensor(mix[:, :, None, None] * bil[None, None], dtype=t.dtype).contiguous()
        elif key.endswith("running_mean") or key.endswith(".bias"):
            out[key] = torch.zeros_like(t)
    return out
