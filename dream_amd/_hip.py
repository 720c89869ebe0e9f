"""ctypes binding of libdream_hip.so (the C ABI declared in include/dream_hip.h).

This is the reference-side binding a DREAM maintainer would add: plain
pointers and sizes, ``hipStream_t`` taken from ``torch.cuda.current_stream()``.  There is NO
fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdream_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dream_hip.h")

CONV_RELU = 1
CONV_UPSAMPLE2X = 2
CONV_OUT_NCHW = 4

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_SZ = _c.c_size_t
_F = _c.c_float
_D = _c.c_double

# name -> (restype, argtypes); must list every function of include/dream_hip.h (checked by
# check_symbols(), which the CPU test-suite runs)
_SIGNATURES = {
    "dream_hip_abi_version": (_I, []),
    "dream_hip_last_error": (_c.c_char_p, []),
    "dream_hip_device_count": (_I, [_c.POINTER(_I)]),
    "dream_hip_device_name": (_I, [_I, _c.c_char_p, _SZ]),
    "dream_hip_stream_create": (_I, [_I, _I, _c.POINTER(_P)]),
    "dream_pack_conv3x3_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dream_unpack_conv3x3_weight": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_pack_conv_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_pack_convT4x4_weight": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_conv3x3_cout_pad": (_SZ, [_I]),
    "dream_conv2d_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv_transpose4x4s2_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv4x4s2_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv2d_wgrad_workspace": (_SZ, [_I, _I, _I, _I, _I, _I, _I]),
    "dream_conv2d_wgrad_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_convT4x4_wgrad_workspace": (_SZ, [_I, _I, _I, _I, _I]),
    "dream_convT4x4_wgrad_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_convT4x4_wgrad_winograd_applies": (_I, [_I, _I]),
    "dream_convT4x4_wgrad_winograd_workspace": (_SZ, [_I, _I, _I, _I, _I]),
    "dream_convT4x4_wgrad_winograd_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dream_upsample_conv3x3_wgrad_from_convT4x4": (_I, [_P, _P, _I, _I, _P]),
    "dream_wgrad_set_variant": (_I, [_I]),
    "dream_wgrad_set_width": (_I, [_I]),
    "dream_convT_wgrad_workspace": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "dream_convT_wgrad_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_unpack_conv_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dream_bn_workspace": (_SZ, [_I]),
    "dream_bn_train_fwd_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _F, _F, _I, _P]),
    "dream_bn_train_bwd_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _P]),
    "dream_channel_sum_nhwc_f32": (_I, [_P, _P, _P, _SZ, _I, _P]),
    "dream_bn_stats_workspace": (_SZ, [_I]),
    "dream_bn_stats_counters": (_I, [_I]),
    "dream_bn_stats_set_pixels_per_row": (_I, [_I]),
    "dream_bn_stats_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _SZ, _I, _P]),
    "dream_bn_apply_ab_nhwc_f32": (_I, [_P, _P, _P, _P, _SZ, _I, _I, _P]),
    "dream_bn_bwd_stats_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _P]),
    "dream_bn_bwd_apply_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _P]),
    "dream_conv1x1_bn_workspace": (_SZ, [_c.c_long, _I]),
    "dream_conv1x1_bn_counters": (_I, [_c.c_long, _I]),
    "dream_conv1x1_bnstats_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _c.c_long, _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "dream_conv3x3_winograd_bn_workspace": (_SZ, [_I, _I, _I, _I]),
    "dream_conv3x3_winograd_bn_counters": (_I, [_I, _I, _I, _I]),
    "dream_conv3x3_winograd_bnstats_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "dream_conv3x3_winograd_bwd_bnmask_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dream_conv1x1_bwd_bnmask_nhwc_f32": (_I, [_P, _P, _P, _P, _c.c_long, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dream_conv1x1_wgrad_pre_nhwc_f32": (_I, [_P, _P, _P, _P, _c.c_long, _I, _I, _I, _P, _P]),
    "dream_maxpool3s2_bwd_nhwc_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dream_maxpool3s2_idx_nhwc_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dream_maxpool3s2_idx_bwd_nhwc_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dream_add_inplace_f32": (_I, [_P, _P, _SZ, _P]),
    "dream_copy_f32": (_I, [_P, _P, _SZ, _P]),
    "dream_copy_chunk_bytes": (_SZ, []),
    "dream_multi_copy_f32": (_I, [_P, _P, _I, _P]),
    "dream_allreduce_sum_f32": (_I, [_I, _c.POINTER(_I), _c.POINTER(_P), _SZ, _c.POINTER(_P)]),
    "dream_allreduce_uses_rccl": (_I, [_I, _c.POINTER(_I)]),
    "dream_add_f32": (_I, [_P, _P, _P, _SZ, _P, _P]),
    "dream_stage_input_nhwc_f32": (_I, [_P, _P, _P] + [_I] * 7 + [_P, _P]),
    "dream_stage_input_bwd_f32": (_I, [_P, _P] + [_I] * 8 + [_P]),
    "dream_conv2d_amax_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv3x3_first_nchw_amax_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_absmax_f32": (_I, [_P, _SZ, _P, _P]),
    "dream_pack_conv_weight_f16x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv2d_f16x3_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_pack_convT4x4_weight_f16x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dream_conv_transpose4x4s2_f16x3_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv_transpose3x3s2_f16x3_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv_f16x3_set_variant": (_I, [_I]),
    "dream_bn_fold_f32": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _I, _P]),
    "dream_im2col_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_maxpool3s2_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_conv3x3_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv1x1_weight_floats": (_SZ, [_I, _I]),
    "dream_conv1x1_set_ksplit": (_I, [_I]),
    "dream_conv1x1_set_rows": (_I, [_I]),
    "dream_conv1x1_wgrad_workspace": (_SZ, [_c.c_long, _I, _I]),
    "dream_conv1x1_wgrad_nhwc_f32": (_I, [_P, _P, _P, _P, _c.c_long, _I, _I, _I, _P]),
    "dream_pack_conv1x1_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "dream_conv1x1_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _c.c_long, _I, _I, _I, _I, _P]),
    "dream_conv1x1_pre_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _c.c_long, _I, _I, _I, _P]),
    "dream_conv3x3_winograd_weight_floats": (_SZ, [_I, _I]),
    "dream_conv3x3_winograd_set_variant": (_I, [_I]),
    "dream_conv3x3_winograd_set_max_workgroups": (_I, [_I]),
    "dream_pack_conv3x3_winograd_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "dream_conv3x3_winograd_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv3x3_winograd4_weight_floats": (_SZ, [_I, _I]),
    "dream_conv3x3_winograd4_set_max_workgroups": (_I, [_I]),
    "dream_conv3x3_winograd4_set_channel_block_pinning": (_I, [_I]),
    "dream_conv3x3_winograd4_set_stagger": (_I, [_I, _I]),
    "dream_pack_conv3x3_winograd4_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "dream_conv3x3_winograd4_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv3x3_winograd4_pool_both_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_pack_job_bytes": (_SZ, []),
    "dream_pack_weights_batched": (_I, [_P, _I, _I, _P]),
    "dream_pack_span_bytes": (_SZ, []),
    "dream_pack_weights_spans": (_I, [_P, _P, _I, _P]),
    "dream_convT4x4_winograd_weight_floats": (_SZ, [_I, _I]),
    "dream_pack_convT4x4_winograd_weight": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "dream_conv4x4s2_winograd_nhwc_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dream_conv_transpose4x4s2_winograd_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_convT4x4_phase_weights": (_I, [_P, _P, _I, _I, _I, _P]),
    "dream_convT4x4_winograd4_weight_floats": (_SZ, [_I, _I]),
    "dream_pack_convT4x4_winograd4_weight": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "dream_conv4x4s2_winograd4_nhwc_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dream_conv_transpose4x4s2_winograd4_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_upsample_conv3x3_weight_as_convT4x4": (_I, [_P, _P, _I, _I, _P]),
    "dream_conv2d_s2_bwd_data_nhwc_f32": (_I, [_P, _P, _P] + [_I] * 9 + [_P]),
    "dream_conv_transpose3x3s2_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv_transpose3x3s2_res_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv3x3_set_variant": (_I, [_I]),
    "dream_conv3x3_num_variants": (_I, []),
    "dream_conv3x3_variant_name": (_c.c_char_p, [_I]),
    "dream_conv3x3_first_nchw_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dream_maxpool2_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_nchw_to_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_nchw_to_nhwc_pad_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dream_nhwc_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_keypoints_from_belief_maps_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _D, _P]),
    "dream_keypoints_from_belief_maps_rule_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _D, _I, _D, _P]),
    "dream_peaks_set_fused": (_I, [_I]),
    "dream_peaks_from_belief_maps_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _D, _P]),
    "dream_gaussian_sigma3_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "dream_softargmax_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "dream_convert_keypoints_f64": (_I, [_P, _P, _P, _I] + [_D] * 8 + [_I, _P]),
    "dream_loss_workspace": (_SZ, [_SZ]),
    "dream_mse_fwd_bwd_f32": (_I, [_P, _P, _P, _P, _P, _SZ, _D, _P]),
    "dream_normalize_u8_hwc_to_chw_f32": (_I, [_P, _P, _I, _I, _I, _c.POINTER(_F), _c.POINTER(_F), _P]),
    "dream_create_belief_maps_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dream_create_belief_maps_f64kps_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dream_smoothl1_fwd_bwd_f32": (_I, [_P, _P, _P, _P, _P, _SZ, _D, _P]),
    "dream_relu_bwd_f32": (_I, [_P, _P, _P, _SZ, _P]),
    "dream_maxpool2_bwd_nhwc_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dream_maxpool2_relu_bwd_nhwc_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dream_upsample2_bwd_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_subsample2_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_col2im4s2_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dream_im2col3s2_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_col2im3s2_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_scatter2_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dream_conv3x3_wgrad_workspace": (_SZ, [_I, _I, _I, _I, _I]),
    "dream_conv3x3_wgrad_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv3x3_wgrad_winograd_workspace": (_SZ, [_I, _I, _I, _I, _I]),
    "dream_conv3x3_wgrad_winograd_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv3x3_wgrad_winograd_fuses_bias": (_I, [_I, _I, _I]),
    "dream_conv3x3_wgrad_winograd_bias_nhwc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dream_conv3x3_wgrad_winograd_set_version": (_I, [_I]),
    "dream_conv3x3_first_wgrad_workspace": (_SZ, [_I, _I, _I, _I, _I]),
    "dream_conv3x3_first_wgrad_f32": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _P]),
    "dream_adam_step_f32": (_I, [_P, _P, _P, _P, _SZ, _F, _F, _F, _F, _I, _P]),
    "dream_sgd_step_f32": (_I, [_P, _P, _SZ, _F, _P]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def header_functions():
    """Function names declared in include/dream_hip.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(dream_[A-Za-z0-9_]+)\s*\(", text)))


def lib():
    """Load (once) and return the ctypes library; raises HipLibraryError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                "libdream_hip.so is not built (%s). Run `python __graft_entry__.py` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check_symbols():
    """Every function the header declares must be exported and bound (no compute is run)."""
    handle = lib()
    declared = header_functions()
    missing = [n for n in declared if not hasattr(handle, n)]
    unbound = [n for n in declared if n not in _SIGNATURES]
    if missing or unbound:
        raise HipLibraryError("missing exports %s / unbound %s" % (missing, unbound))
    if handle.dream_hip_abi_version() != 2:
        raise HipLibraryError("ABI version mismatch")
    return declared


def call(name, *args):
    """Invoke a status-returning entry point; non-zero -> RuntimeError with the library's message."""
    handle = lib()
    rc = getattr(handle, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, handle.dream_hip_last_error().decode()))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  The tensor must be contiguous fp32/int32/fp64
    on a GPU: a CPU tensor here means the caller tried to run the HIP path without a device."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("dream_amd: tensor is on %s; the HIP path needs a GPU tensor (no CPU fallback)"
                           % t.device)
    if not t.is_contiguous():
        raise RuntimeError("dream_amd: non-contiguous tensor passed to the HIP library")
    return t.data_ptr()


def device_tensor(t):
    """The reference's functions accept host or device tensors; the HIP path needs device memory."""
    import torch
    if t.is_cuda:
        return t
    if not torch.cuda.is_available():
        raise RuntimeError("dream_amd: this operation runs on the GPU only (no CPU fallback) and no GPU is visible")
    return t.cuda()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def stream_on(device):
    """HIP stream the calling thread currently uses on ``device`` (a torch.device of type cuda)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


_own_streams = {}
_own_lock = None


def own_stream(device, role, priority=0):
    """The process's own HIP stream for ``role`` on ``device`` (a torch.device of type cuda), as a torch stream: created once by
    dream_hip_stream_create and kept for the life of the process.  torch.cuda.Stream() draws from a pool of 32 streams per device
    and priority, round-robin: in a long-running process the 33rd request returns the first stream again, and a hipGraph capture
    begun on such an alias swallows whatever another thread launches on "its" stream.  Roles: "leaves" (the weight-gradient stream
    of training steps), "exchange" (gradient exchange), "capture" / "capture-leaves" (hipGraph captures; taken one at a time,
    process-wide), "warmup"."""
    import threading
    import torch
    global _own_lock
    if _own_lock is None:
        _own_lock = threading.Lock()
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, role, priority)
    with _own_lock:
        st = _own_streams.get(key)
        if st is None:
            handle = _c.c_void_p()
            call("dream_hip_stream_create", int(index), int(priority), _c.byref(handle))
            st = _own_streams[key] = torch.cuda.ExternalStream(handle.value, device=torch.device("cuda", index))
    return st


def cout_pad(cout):
    return int(lib().dream_conv3x3_cout_pad(cout))
