"""CPU suite: the product's HIP kernel sources, compiled UNCHANGED against the SIMT emulator
(tests/emu), driven through the product's host code (dream_amd.ops / models / network) and checked
against the oracle and the reference's golden outputs.  This validates index arithmetic, LDS layouts,
barrier placement, the documented MFMA lane layout and all host logic without a GPU; the same checks run
on the real device in the -m gpu suite."""
import os

import pytest
import torch

import cases
from dream_amd import ops
import parity_checks as pc
from emu_util import emulated_hip


@pytest.fixture(scope="module")
def emu():
    with emulated_hip() as lib:
        yield lib


@pytest.mark.parametrize("variant", range(11))
def test_conv_variants(emu, variant):
    emu.dream_conv3x3_set_variant(variant)
    try:
        pc.check_conv("cpu", 1, 7, 9, 32, 40, 1, seed=variant)
        pc.check_conv("cpu", 2, 12, 20, 64, 7, 4, seed=variant)          # NCHW store, Cout < 32
        pc.check_conv("cpu", 1, 6, 8, 32, 64, 3, seed=variant)           # fused upsample + ReLU
    finally:
        emu.dream_conv3x3_set_variant(-1)


@pytest.mark.parametrize("variant", [4, 8])              # 64- / 128-channel workgroups
def test_conv_winograd(emu, variant):
    emu.dream_conv3x3_winograd_set_variant(variant)
    errs = [pc.check_conv_winograd("cpu", 1, 8, 8, 32, 16),                                   # one workgroup, ragged cout
            pc.check_conv_winograd("cpu", 2, 13, 25, 32, 64, ops.CONV_RELU, seed=1),           # odd extents: half tiles
            pc.check_conv_winograd("cpu", 1, 25, 25, 48, 96, ops.CONV_RELU, seed=2),           # tiles spanning images / rows
            pc.check_conv_winograd("cpu", 3, 5, 3, 32, 7, 0, seed=3),                          # images smaller than a block
            pc.check_conv_winograd("cpu", 2, 12, 20, 32, 80, ops.CONV_RELU | ops.CONV_POOL2, seed=4),
            pc.check_conv_winograd("cpu", 2, 13, 9, 32, 64, ops.CONV_RELU | ops.CONV_POOL2, seed=8),   # odd extents: floor
            pc.check_conv_winograd("cpu", 1, 10, 14, 64, 32, ops.CONV_RELU, seed=5, with_scale=True, residual="add"),
            pc.check_conv_winograd("cpu", 1, 9, 11, 32, 48, ops.CONV_RELUMASK, seed=6, residual="mask"),
            pc.check_conv_winograd("cpu", 2, 7, 9, 32, 64, 0, seed=7, mode=1),
            # 25 tile blocks on a grid of 8 workgroups: every workgroup walks over 3-4 blocks (next block's first chunk
            # prefetched during the last chunk of the current one); 3 chunks: the V buffer parity flips from block to block
            pc.check_conv_winograd("cpu", 2, 40, 40, 32, 64, ops.CONV_RELU | ops.CONV_POOL2, seed=9, max_workgroups=(8,)),
            pc.check_conv_winograd("cpu", 1, 56, 55, 48, 32, ops.CONV_RELUMASK, seed=10, residual="mask", max_workgroups=(8,))]
    emu.dream_conv3x3_winograd_set_variant(0)
    print("winograd max rel err", max(errs))


def test_conv_winograd4(emu):
    # up to 64 output channels: the narrow workgroup shape (4 wavefronts, 8-channel chunks, two workgroups per CU)
    errs = [pc.check_conv_winograd4("cpu", 1, 8, 8, 32, 16, max_workgroups=()),                          # one workgroup, ragged cout
            pc.check_conv_winograd4("cpu", 2, 13, 25, 32, 64, ops.CONV_RELU, seed=1),                     # extents not multiples of 4
            pc.check_conv_winograd4("cpu", 3, 5, 3, 48, 7, 0, seed=3, max_workgroups=()),                 # images smaller than a tile block, 6 chunks
            pc.check_conv_winograd4("cpu", 2, 13, 9, 16, 64, ops.CONV_RELU | ops.CONV_POOL2, seed=8, max_workgroups=()),   # odd extents: floor; 2 chunks
            pc.check_conv_winograd4("cpu", 1, 10, 14, 64, 32, ops.CONV_RELU, seed=5, with_scale=True, residual="add", max_workgroups=()),
            pc.check_conv_winograd4("cpu", 1, 9, 11, 32, 48, ops.CONV_RELUMASK, seed=6, residual="mask", max_workgroups=()),
            pc.check_conv_winograd4("cpu", 2, 7, 9, 32, 64, 0, seed=7, mode=1, max_workgroups=()),
            # 25 tile blocks on a grid of 8 workgroups: several blocks per workgroup, the next block's first chunk transformed
            # during the last chunk of the current one
            pc.check_conv_winograd4("cpu", 2, 40, 40, 32, 64, ops.CONV_RELU | ops.CONV_POOL2, seed=9)]
    # more than 64: the wide shape (8 wavefronts, 16-channel chunks)
    errs += [pc.check_conv_winograd4("cpu", 1, 8, 8, 32, 80, max_workgroups=()),
             pc.check_conv_winograd4("cpu", 2, 13, 25, 32, 128, ops.CONV_RELU, seed=1),
             pc.check_conv_winograd4("cpu", 3, 5, 3, 64, 71, 0, seed=3, max_workgroups=()),                # 4 chunks
             pc.check_conv_winograd4("cpu", 2, 12, 20, 32, 144, ops.CONV_RELU | ops.CONV_POOL2, seed=4),   # two channel blocks, fused pool
             pc.check_conv_winograd4("cpu", 1, 10, 14, 64, 96, ops.CONV_RELU, seed=5, with_scale=True, residual="add", max_workgroups=()),
             pc.check_conv_winograd4("cpu", 1, 9, 11, 32, 112, ops.CONV_RELUMASK, seed=6, residual="mask", max_workgroups=()),
             pc.check_conv_winograd4("cpu", 2, 7, 9, 32, 128, 0, seed=7, mode=1, max_workgroups=()),
             pc.check_conv_winograd4("cpu", 2, 40, 40, 32, 128, ops.CONV_RELU | ops.CONV_POOL2, seed=9)]
    print("winograd F(4x4) max rel err", max(errs))
    # MODE 4: un-pooled + pooled tensor from one launch (training forward), narrow and wide shape, odd extents, several blocks per workgroup
    for (b, h, w, cin, cout) in [(2, 13, 9, 16, 64), (1, 8, 12, 32, 48), (2, 12, 20, 32, 144), (2, 25, 25, 32, 128), (2, 40, 40, 32, 64)]:
        pc.check_conv_winograd4_pool_both("cpu", b, h, w, cin, cout, seed=b + h)


def test_conv_transpose4x4_winograd(emu, monkeypatch):
    # small grids run on four-wavefront workgroups since round 5 (csrc/conv_wino.hip small_grid_nw4): both forms on the same cases
    monkeypatch.setenv("DREAM_WINO_SMALL_GRID", "0")
    errs8 = [pc.check_convT4x4_winograd("cpu", 1, 6, 6, 32, 128, max_workgroups=()),
             pc.check_convT4x4_winograd("cpu", 2, 13, 9, 48, 96, ops.CONV_RELU, seed=1, with_scale=True)]
    monkeypatch.setenv("DREAM_WINO_SMALL_GRID", "1")
    print("winograd convT (eight-wavefront workgroups forced) max rel err", max(errs8))
    errs = [pc.check_convT4x4_winograd("cpu", 1, 6, 6, 32, 128, max_workgroups=()),                        # one block per phase
            pc.check_convT4x4_winograd("cpu", 2, 13, 9, 48, 96, ops.CONV_RELU, seed=1, with_scale=True),    # odd extents, 3 chunks, ragged cout
            pc.check_convT4x4_winograd("cpu", 1, 26, 26, 32, 256, ops.CONV_RELU, seed=2)]                   # two channel blocks, several tile blocks per workgroup
    print("convT4x4 winograd max rel err", max(errs))
    errs4 = [pc.check_convT4x4_winograd("cpu", 1, 8, 8, 32, 128, max_workgroups=(), tile=4),                      # one block per phase
             pc.check_convT4x4_winograd("cpu", 2, 13, 9, 64, 96, ops.CONV_RELU, seed=1, with_scale=True, max_workgroups=(), tile=4),   # odd extents, 4 chunks, ragged cout
             pc.check_convT4x4_winograd("cpu", 1, 26, 26, 32, 256, ops.CONV_RELU, seed=2, tile=4)]                 # two channel blocks, several tile blocks per workgroup
    print("convT4x4 winograd F(4x4) max rel err", max(errs4))
    berrs = [pc.check_conv4x4s2_winograd("cpu", 1, 6, 6, 128, 32),                 # data gradient: one block per phase
             pc.check_conv4x4s2_winograd("cpu", 2, 13, 9, 96, 48, seed=1),          # odd extents, 3 chunks, ragged channels
             pc.check_conv4x4s2_winograd("cpu", 1, 20, 22, 256, 32, seed=2)]        # two channel blocks
    print("conv4x4s2 (convT data gradient) winograd max rel err", max(berrs))
    berrs4 = [pc.check_conv4x4s2_winograd("cpu", 1, 8, 8, 128, 32, tile=4),                 # one block per phase
              pc.check_conv4x4s2_winograd("cpu", 2, 13, 9, 96, 64, seed=1, tile=4),          # odd extents, 4 chunks, ragged channels
              pc.check_conv4x4s2_winograd("cpu", 1, 20, 22, 256, 32, seed=2, tile=4)]        # two channel blocks
    print("conv4x4s2 (convT data gradient) winograd F(4x4) max rel err", max(berrs4))


def test_conv1x1_gemm(emu):
    errs = [pc.check_conv1x1("cpu", 1, 8, 8, 64, 64),                                             # one tile
            pc.check_conv1x1("cpu", 2, 5, 7, 128, 36, ops.CONV_RELU, seed=1, with_scale=True),    # ragged rows and channels
            pc.check_conv1x1("cpu", 1, 13, 13, 256, 192, ops.CONV_RELU, seed=2, residual=True),   # several tiles, K split 4
            pc.check_conv1x1("cpu", 3, 4, 3, 64, 128, 0, seed=3, mode=1, residual=True),          # data-gradient operator
            pc.check_conv1x1("cpu", 1, 9, 9, 160, 64, ops.CONV_RELU, seed=4)]                     # K = 160 (the stem's im2col)
    print("conv1x1 gemm max rel err", max(errs))
    werrs = [pc.check_conv1x1_wgrad("cpu", 1, 8, 8, 64, 64),                     # one tile, one split
             pc.check_conv1x1_wgrad("cpu", 2, 5, 7, 128, 36, seed=1),            # ragged position count and output channels
             pc.check_conv1x1_wgrad("cpu", 1, 33, 31, 64, 128, seed=2),          # several splits, tail beyond the last position
             pc.check_conv1x1_wgrad("cpu", 3, 4, 3, 192, 64, seed=3, pad_dy=16)] # padded gradient tensor
    print("conv1x1 wgrad max err / sum|terms|", max(werrs))


def test_upsample_conv_wgrad_on_the_transposed_conv_form(emu, monkeypatch):
    """DREAM_UPS_WGRAD=convT9 (opt-in): nn.Upsample(2) + conv3x3's weight gradient as the transposed conv's nine-position gradient + the
    4x4 -> 3x3 tap sums, against fp64 and the default sixteen-position kernel with the fused upsample."""
    monkeypatch.setattr(ops, "UPS_WGRAD_AS_CONVT", True)
    errs = [pc.check_wgrad_winograd("cpu", 2, 12, 10, 64, 64, seed=9, ups=True),
            pc.check_wgrad_winograd("cpu", 1, 26, 26, 128, 64, seed=10, ups=True)]
    print("upsample + conv3x3 wgrad via convT9: max err / sum|terms|", max(errs))


def test_convT4x4_wgrad_winograd(emu):
    """The transposed conv's weight gradient on nine of the sixteen Winograd positions (round 6)."""
    errs = [pc.check_convT4x4_wgrad_winograd("cpu", 1, 8, 8, 64, 64),                # one block, one split
            pc.check_convT4x4_wgrad_winograd("cpu", 2, 13, 9, 64, 64, seed=1),       # odd phase extents (half tiles), several stages
            pc.check_convT4x4_wgrad_winograd("cpu", 1, 5, 3, 128, 64, seed=2),       # fewer tiles than a stage, two input-channel blocks
            pc.check_convT4x4_wgrad_winograd("cpu", 1, 26, 26, 64, 128, seed=3)]     # two output-channel blocks, several splits
    print("convT4x4 wgrad (F(2x2,2x2), 9 positions) max err / sum|terms|", max(errs))


def test_wgrad_winograd(emu):
    errs = [pc.check_wgrad_winograd("cpu", 1, 8, 8, 64, 16),                    # one split, interior + border tiles
            pc.check_wgrad_winograd("cpu", 2, 13, 9, 64, 32, seed=1),            # odd extents (half tiles), several images
            pc.check_wgrad_winograd("cpu", 3, 6, 10, 128, 48, seed=2),           # two input-channel groups, ragged output group
            pc.check_wgrad_winograd("cpu", 1, 25, 25, 64, 64, seed=3),           # several splits
            pc.check_wgrad_winograd("cpu", 2, 5, 3, 64, 16, seed=4, pad_dy=16),  # images smaller than a k-step, padded dy
            # 64-multiples of output channels: the LDS-staged kernel (stages of 8 tiles, operands shared by the workgroup)
            pc.check_wgrad_winograd("cpu", 2, 13, 9, 64, 64, seed=5),            # odd extents, several stages
            pc.check_wgrad_winograd("cpu", 3, 6, 10, 128, 64, seed=6),           # two input-channel blocks
            pc.check_wgrad_winograd("cpu", 1, 25, 25, 64, 128, seed=7),          # two output-channel blocks, several splits
            pc.check_wgrad_winograd("cpu", 2, 5, 3, 64, 64, seed=8, pad_dy=16),  # fewer tiles than a stage, padded dy
            pc.check_wgrad_winograd("cpu", 2, 12, 10, 64, 64, seed=9, ups=True),  # conv after a nearest x2 upsample (fused)
            pc.check_wgrad_winograd("cpu", 1, 26, 26, 128, 64, seed=10, ups=True)]
    print("winograd wgrad max err / sum|terms|", max(errs))


def test_conv_heuristic_and_odd_shapes(emu):
    pc.check_conv("cpu", 1, 25, 25, 64, 128, 1)       # 5x25 tiles
    pc.check_conv("cpu", 3, 5, 3, 16, 16, 0)          # KC=16 fallback, image smaller than a tile
    pc.check_conv("cpu", 1, 13, 31, 48, 32, 1)
    pc.check_conv_transpose("cpu", 1, 5, 7, 32, 48)


def test_general_conv_resnet_ops(emu):
    pc.check_conv2d_general("cpu", 2, 9, 11, 64, 48, 1, 1)
    pc.check_conv2d_general("cpu", 1, 13, 13, 32, 128, 1, 2)      # strided 1x1 (Bottleneck downsample)
    pc.check_conv2d_general("cpu", 2, 12, 10, 32, 64, 3, 2)       # strided 3x3 (big-patch variant)
    pc.check_conv2d_general("cpu", 1, 13, 25, 64, 32, 3, 2)
    pc.check_conv_transpose4x4("cpu", 1, 5, 6, 32, 48)
    pc.check_conv_transpose4x4("cpu", 2, 13, 13, 64, 32)
    pc.check_resnet_stem("cpu", 2, 30, 37)


def test_resnet_h_inference_golden(emu):
    pc.check_model_inference("cpu", "resnet_h", (2, 64, 96))


def test_first_conv_pool_layouts(emu):
    pc.check_first_conv("cpu", 1, 16, 16)
    pc.check_first_conv("cpu", 2, 21, 37)
    pc.check_pool_and_layouts("cpu")


@pytest.mark.parametrize("name", [n for n, (m, _) in cases.peak_cases().items() if m.shape[1] * m.shape[2] <= 50000])
def test_peaks_bit_exact(emu, name):
    pc.check_peaks_case("cpu", name)


def test_gaussian_staged_strips_ragged_sizes(emu):
    """The LDS-staged Gaussian passes (64-column strips, 16 / 4 output rows per workgroup, reflection at staging time) on maps that
    are smaller than the filter radius, one pixel wide or tall, and not multiples of the strip sizes: scipy's bits."""
    import numpy as np
    import torch
    from oracle import peaks as op
    rng = np.random.default_rng(0)
    for (n, h, w) in [(2, 5, 7), (1, 13, 70), (2, 40, 130), (1, 17, 64), (1, 16, 65), (3, 1, 1), (1, 2, 200), (1, 33, 3)]:
        m = rng.standard_normal((n, h, w)).astype(np.float32)
        sm = ops.gaussian_sigma3(torch.from_numpy(m)).numpy()
        for i in range(n):
            assert np.array_equal(sm[i], op.gaussian_filter_sigma3(m[i])), (n, h, w, i)


def test_peak_rule_settings(emu):
    pc.check_peak_rule_settings("cpu")


def test_peaks_fused_row_pass_and_scan_same_bits(emu):
    pc.check_peaks_fused_equals_three_kernels("cpu", sizes=((1, 7, 5, 9), (2, 3, 37, 45), (1, 2, 64, 64), (1, 2, 30, 71)))


def test_peaks_api_reference_kat(emu):
    pc.check_peaks_api("cpu")


def test_softargmax(emu):
    pc.check_softargmax("cpu")


def test_backward_ops(emu):
    pc.check_backward_ops("cpu")


def test_vgg_q_inference_golden(emu):
    pc.check_model_inference("cpu", "vgg_q", (1, 50, 75))


def test_vgg_f_inference_golden(emu):
    pc.check_model_inference("cpu", "vgg_f", (2, 64, 80))


def test_train_step_golden(emu):
    pc.check_train_steps("cpu", "adam", steps=1)


def test_resnet_training_ops(emu):
    pc.check_resnet_training_ops("cpu")


def test_resnet_h_train_step(emu):
    pc.check_resnet_train_step("cpu", "resnet_h", (2, 64, 64))


def test_bn_fused_ops(emu):
    """Round 4: BatchNorm statistics finished inside the producing launch, BN + ReLU in the consumer's loader, backward reductions
    in the data-gradient epilogue (csrc/bn.hip, csrc/gemm1x1.hip)."""
    pc.check_bn_fused_ops("cpu")


def test_conv3x3_bn_fused_ops(emu):
    """The 3x3 convs' BatchNorm in the Winograd F(2x2) kernel's epilogue (csrc/conv_wino.hip WINO_STAT): forward statistics, backward
    mask + reductions; one-level trees under the suite's grid cap, then the chip's grid (24 producer rows: both levels of the tree,
    idle workgroups arriving with rows of zeros)."""
    pc.check_conv3x3_bn_fused("cpu")
    emu.dream_conv3x3_winograd_set_max_workgroups(0)
    try:
        pc.check_conv3x3_bn_fused("cpu", cases=[(4, 25, 25, 64, 64), (1, 7, 9, 64, 128)])
    finally:
        emu.dream_conv3x3_winograd_set_max_workgroups(16)


def test_resnet_h_train_step_bn_in_the_3x3_kernels(emu, monkeypatch):
    """DREAM_BN_FUSION_3X3=1 (opt-in): conv2's statistics and bn1's backward reductions ride in the Winograd kernel -- the step still
    meets the reference goldens, and the launches it removes are really gone."""
    monkeypatch.setenv("DREAM_BN_FUSION_3X3", "1")
    seen = []
    real_stats, real_fwd, real_bwd = ops.bn_stats, ops.conv3x3_winograd_bn, ops.conv3x3_winograd_bwd_bnmask
    monkeypatch.setattr(ops, "bn_stats", lambda *a, **k: (seen.append("bn_stats"), real_stats(*a, **k))[1])
    monkeypatch.setattr(ops, "conv3x3_winograd_bn", lambda *a, **k: (seen.append("fwd"), real_fwd(*a, **k))[1])
    monkeypatch.setattr(ops, "conv3x3_winograd_bwd_bnmask", lambda *a, **k: (seen.append("bwd"), real_bwd(*a, **k))[1])
    pc.check_resnet_train_step("cpu", "resnet_h", (2, 64, 64))
    # ResNet-101: 33 Bottlenecks; the three stride-2 conv2's (layers 2-4, first block) are not Winograd convs
    assert seen.count("fwd") >= 30 and seen.count("bwd") >= 30, (seen.count("fwd"), seen.count("bwd"))
    assert seen.count("bn_stats") <= 12, seen.count("bn_stats")


def test_resnet_h_train_step_three_launch_batchnorm(emu, monkeypatch):
    """The round-1..3 BatchNorm kernels (DREAM_BN_FUSION=0) stay selectable and correct."""
    monkeypatch.setenv("DREAM_BN_FUSION", "0")
    pc.check_resnet_train_step("cpu", "resnet_h", (2, 64, 64))


def test_split_precision_conv(emu):
    for v in range(6):
        emu.dream_conv_f16x3_set_variant(v)
        try:
            pc.check_conv_f16x3("cpu", 1, 7, 9, 32, 40, 3, 1, seed=v)
            pc.check_conv_f16x3("cpu", 2, 12, 20, 64, 7, 3, 4, x_scale=300.0, w_scale=1e-3, seed=v)
            pc.check_conv_f16x3("cpu", 1, 6, 8, 32, 64, 3, 3, x_scale=1e-3, w_scale=5.0, seed=v)
            pc.check_conv_f16x3("cpu", 2, 9, 11, 64, 48, 1, 0, seed=v)
        finally:
            emu.dream_conv_f16x3_set_variant(-1)


def test_vgg_q_inference_golden_split_precision(emu):
    pc.check_model_inference("cpu", "vgg_q", (1, 50, 75), precision="fp16x3")


def test_vgg_f_train_step(emu):
    pc.check_vgg_train_grads("cpu", "vgg_f", (2, 32, 48))


def test_fused_maxpool_epilogue(emu):
    for v in (0, 1, 3, 8):
        emu.dream_conv3x3_set_variant(v)
        try:
            pc.check_conv("cpu", 2, 12, 20, 32, 48, 1 | 16, seed=v)
            pc.check_conv("cpu", 1, 13, 9, 64, 32, 1 | 16, seed=v)        # odd extent: floor pooling drops the last row/col
        finally:
            emu.dream_conv3x3_set_variant(-1)
    for v in (0, 1, 2, 4, 5):
        emu.dream_conv_f16x3_set_variant(v)
        try:
            pc.check_conv_f16x3("cpu", 2, 12, 20, 32, 48, 3, 1 | 16, seed=v)
            pc.check_conv_f16x3("cpu", 1, 13, 9, 64, 32, 3, 1 | 16, seed=v)
        finally:
            emu.dream_conv_f16x3_set_variant(-1)


def test_on_device_dataprep(emu):
    pc.check_dataprep("cpu")


def test_resnet_split_precision(emu):
    pc.check_conv_transpose4x4_f16x3("cpu", 1, 5, 6, 32, 48)
    if os.environ.get("DREAM_EMU_FULL", "0") == "1":       # ~50 s under the emulator; the GPU suite runs it always
        pc.check_model_inference("cpu", "resnet_h", (2, 64, 96), precision="fp16x3")


_FULL = os.environ.get("DREAM_EMU_FULL", "0") == "1"    # the emulator is ~1e4x slower than the GPU: the default CPU suite
                                                        # runs two inference cases, DREAM_EMU_FULL=1 all of them + training


@pytest.mark.parametrize("name", sorted(cases.VARIANT_CASES) if _FULL else ["vgg_f_ms2_skip", "vgg_ms2"])
def test_hourglass_variants(emu, name):
    pc.check_variant("cpu", name, train=_FULL)


@pytest.mark.skipif(not _FULL, reason="set DREAM_EMU_FULL=1 (several minutes under the emulator); covered on the GPU")
def test_hourglass_variants_split_precision(emu):
    pc.check_variant("cpu", "vgg_f_ms2_skip", precision="fp16x3")


def test_keypoint_frame_conversions(emu):
    pc.check_keypoint_conversions("cpu")


def test_conv_transpose3x3_subpixel(emu):
    pc.check_conv_transpose3x3("cpu", 2, 5, 7, 32, 48)
    pc.check_conv_transpose3x3("cpu", 1, 9, 4, 64, 16, relu=False, seed=3)


def test_upsample_conv_as_transposed_conv(emu):
    pc.check_upsample_conv_as_convT("cpu", 2, 5, 7, 32, 48)
    pc.check_upsample_conv_as_convT("cpu", 1, 3, 9, 64, 16, relu=False, seed=4)


def test_randomised_conv_geometries(emu):
    """Seeded sweep over ragged extents (down to 1 pixel), channel counts that are not tile multiples and every fusion
    flag, for the conv, the two transposed-conv forms and the weight gradient -- all against torch."""
    import numpy as np
    from dream_amd import ops
    rs = np.random.RandomState(2026)
    for case in range(14):
        b = int(rs.randint(1, 3))
        h, w = int(rs.randint(1, 19)), int(rs.randint(1, 23))
        cin = int(rs.choice([16, 32, 48, 80]))
        cout = int(rs.choice([7, 16, 33, 64, 130]))
        flags = int(rs.choice([0, ops.CONV_RELU, ops.CONV_OUT_NCHW, ops.CONV_RELU | ops.CONV_OUT_NCHW]))
        pc.check_conv("cpu", b, h, w, cin, cout, flags, seed=case)
        he, we = 2 * int(rs.randint(1, 8)), 2 * int(rs.randint(1, 9))
        pc.check_conv("cpu", b, he, we, cin, cout, ops.CONV_RELU | ops.CONV_POOL2, seed=case)
        pc.check_conv("cpu", b, he, we, cin, cout, ops.CONV_UPSAMPLE2X | (flags & ops.CONV_RELU), seed=case)
        pc.check_conv_transpose3x3("cpu", b, h, w, cin, cout, relu=bool(flags & ops.CONV_RELU), seed=case)
        pc.check_upsample_conv_as_convT("cpu", b, h, w, cin, cout, relu=bool(flags & ops.CONV_RELU), seed=case)
        co4 = int(rs.choice([8, 16, 64, 132]))               # weight gradients need channel counts that are multiples of 4
        pc.check_wgrad("cpu", b, h, w, cin, co4, k=int(rs.choice([1, 3])), stride=int(rs.choice([1, 2])), seed=case)


def test_batched_weight_packing(emu):
    """dream_pack_weights_batched (one launch, a job table) writes exactly what the one-tensor pack entry points write."""
    import torch
    g = torch.Generator().manual_seed(5)
    jobs, singles = [], []
    for kind, fn, cout, cin, mode in [(ops.PACK_CONV1X1, ops.pack_conv1x1_weight, 64, 96, 0), (ops.PACK_CONV1X1, ops.pack_conv1x1_weight, 64, 40, 1),
                                      (ops.PACK_WINOGRAD2, ops.pack_weight_winograd, 24, 32, 0), (ops.PACK_WINOGRAD2, ops.pack_weight_winograd, 48, 16, 1),
                                      (ops.PACK_WINOGRAD4, ops.pack_weight_winograd4, 20, 32, 0), (ops.PACK_WINOGRAD4, ops.pack_weight_winograd4, 64, 48, 1)]:
        k = 1 if kind == ops.PACK_CONV1X1 else 3
        w = torch.randn(cout, cin, k, k, generator=g)
        with ops.record_packs() as descs:
            ref = fn(w, mode)[0]
        assert len(descs) == 1 and descs[0][0] == kind
        out = ref.clone()
        n_body = out.numel() if kind == ops.PACK_CONV1X1 else out.numel() - (6 if kind == ops.PACK_WINOGRAD2 else 4) * ((((cout if mode == 0 else cin) + 127) // 128) * 128) * 16
        out[:n_body] = float("nan")                           # the batched kernel must rewrite the body (not the zero tail)
        jobs.append((kind, w, out, cout, cin, mode))
        singles.append(ref)
    # the decoder's transposed convs: four jobs (one per output phase) straight from wT, forward and data-gradient operators,
    # F(2x2) and F(4x4) -- against the two-step one-tensor path (materialised zero-padded 3x3 kernels, then the conv's pack)
    for kind, fn, cin_t, cout_t, mode in [(ops.PACK_CONVT_WINOGRAD2, ops.pack_convT4x4_winograd_weight, 32, 48, 0),
                                         (ops.PACK_CONVT_WINOGRAD2, ops.pack_convT4x4_winograd_weight, 48, 32, 1),
                                         (ops.PACK_CONVT_WINOGRAD4, ops.pack_convT4x4_winograd4_weight, 32, 160, 0),
                                         (ops.PACK_CONVT_WINOGRAD4, ops.pack_convT4x4_winograd4_weight, 160, 32, 1)]:
        wT = torch.randn(cin_t, cout_t, 4, 4, generator=g)
        with ops.record_packs() as descs:
            ref = fn(wT, mode)[0]
        assert len(descs) == 4 and all(d[0] == kind for d in descs)
        out = ref.clone()
        per = out.numel() // 4
        rows = cout_t if mode == 0 else cin_t
        k_, pad_ = (16, (rows + 127) // 128 * 128) if (kind == ops.PACK_CONVT_WINOGRAD2 or rows > 64) else (8, 64)
        ahead = 6 if kind == ops.PACK_CONVT_WINOGRAD2 else (6 if rows > 64 else 16)
        for ph, d in enumerate(descs):
            body = per - ahead * pad_ * k_
            out[ph * per:ph * per + body] = float("nan")          # the batched kernel rewrites the body; the zero tail stays
            jobs.append((kind, wT, out[ph * per:(ph + 1) * per], d[3], d[4], d[5]))
            singles.append(ref[ph * per:(ph + 1) * per])
    table = ops.pack_job_table(jobs, "cpu")
    spans, nspans = ops.pack_span_table(jobs, "cpu", floats_per_workgroup=1 << 12)          # workgroups by job size (round 6): same bits
    assert nspans > len(jobs)
    poisoned = [j[2].clone() for j in jobs]
    ops.pack_weights_spans(table, spans, nspans)
    for (kind, w, out, cout, cin, mode), ref, before in zip(jobs, singles, poisoned):
        assert torch.equal(out, ref), (kind, cout, cin, mode)
        out.copy_(before)
    ops.pack_weights_batched(table, len(jobs), workgroups_per_job=3)
    for (kind, w, out, cout, cin, mode), ref in zip(jobs, singles):
        assert torch.equal(out, ref), (kind, cout, cin, mode)


def test_winograd4_channel_blocks_pinned_to_xcds_same_bits(emu):
    """csrc/conv_wino4.hip, Wino4Params::ymap: the 1-D grid whose XCD k owns output-channel block k % ny gives the bits of the
    (tile blocks) x (channel blocks) grid -- 67 tile blocks on 64 emulated workgroups, two and four channel blocks."""
    from dream_amd import _hip
    torch.manual_seed(0)
    for cout in (256, 512):
        x = torch.randn(2, 92, 92, 32).relu_()
        w = torch.randn(cout, 32, 3, 3) * 0.1
        u4, rows = ops.pack_weight_winograd4(w, 0)
        outs = []
        _hip.call("dream_conv3x3_winograd4_set_max_workgroups", 64)
        try:
            for pin in (0, 1):
                _hip.call("dream_conv3x3_winograd4_set_channel_block_pinning", pin)
                outs.append(ops.conv3x3_winograd4(x, u4, cout, None, None, None, ops.CONV_RELU))
        finally:
            _hip.call("dream_conv3x3_winograd4_set_channel_block_pinning", -1)
            _hip.call("dream_conv3x3_winograd4_set_max_workgroups", 0)
        assert torch.equal(outs[0], outs[1])
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).relu().permute(0, 2, 3, 1)
        assert float((outs[1].double() - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("algorithm", ["winograd", "direct"])
def test_skip_connections_fold_into_the_producing_conv(emu, monkeypatch, algorithm):
    """K13 (dream/models.py:774-799): in inference the skip-connection sums x + x_0_k_d are made by the producing conv's epilogue
    (DREAM_CONV_RES_AFTER_RELU) -- no stand-alone add launch -- and the reference's golden maps still hold; training keeps the add."""
    monkeypatch.setenv("DREAM_CONV_ALGORITHM", algorithm)
    adds = []
    real = ops.add
    monkeypatch.setattr(ops, "add", lambda *a, **k: (adds.append(1), real(*a, **k))[1])
    for name in ("vgg_q_skip", "vgg_f_skip"):
        pc.check_variant("cpu", name, train=False)
    assert not adds


def test_multi_copy_gathers_many_tensors_with_one_launch(emu):
    """ops.MultiCopyPlan (dream_multi_copy_f32): the optimizer's gather of per-parameter gradients into its flat buffer -- aligned and
    unaligned sources, tensors longer than one 64 K chunk, the padding between the views untouched, repeated calls (the pointer ring)."""
    torch.manual_seed(0)
    sizes = [1, 3, 64, 65537, 200000, 7, 131072, 5]
    flat = torch.zeros(sum((n + 63) // 64 * 64 for n in sizes))
    views, o = [], 0
    for n in sizes:
        views.append(flat[o:o + n])
        o += (n + 63) // 64 * 64
    plan = ops.MultiCopyPlan(views)
    assert plan.nchunks == sum((n + 65535) // 65536 for n in sizes)
    for rep in range(6):
        srcs = [torch.randn(n + 1)[1:] if (i + rep) % 2 else torch.randn(n) for i, n in enumerate(sizes)]    # every other source 4 bytes off
        assert plan.matches(srcs)
        plan.run(srcs)
        for v, s_ in zip(views, srcs):
            assert torch.equal(v, s_)
        o = 0
        for n in sizes:                                                   # the padding between the views stays zero
            pad = (n + 63) // 64 * 64
            assert float(flat[o + n:o + pad].abs().sum()) == 0.0
            o += pad
    assert not plan.matches(srcs[:-1]) and not plan.matches([t.double() for t in srcs])
