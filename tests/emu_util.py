"""TEST INFRASTRUCTURE: run dream_amd's host code against the SIMT-emulated kernels (tests/emu) on CPU
tensors.  Only the test-suite does this monkeypatching; the product binding (dream_amd/_hip.py) has no
such switch and refuses CPU tensors."""
import contextlib
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


def load_emulated_lib():
    import build_emu
    from dream_amd import _hip
    handle = ctypes.CDLL(build_emu.build())
    for name, (res, args) in _hip._SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = res, args
    return handle


@contextlib.contextmanager
def emulated_hip():
    from dream_amd import _hip, ops
    handle = load_emulated_lib()
    saved = (_hip._lib, _hip.ptr, _hip.stream, ops.ptr, ops.stream, _hip.device_tensor, _hip.stream_on, ops.stream_on)

    def cpu_ptr(t):
        if t is None:
            return None
        assert t.is_contiguous() and not t.is_cuda
        return t.data_ptr()

    # The persistent Winograd kernels size their grid for the chip (256 / 512 workgroups of 256-512 threads) whatever the problem; the
    # emulator pays ~0.1 s per such launch in fiber start-ups for workgroups that find no tile block.  Their own test hook caps the
    # grid (same tile-block walk, same bits -- asserted by the capped / uncapped kernel tests): the whole CPU suite runs capped.
    handle.dream_conv3x3_winograd_set_max_workgroups(16)
    handle.dream_conv3x3_winograd4_set_max_workgroups(16)
    _hip._lib = handle
    _hip.ptr = ops.ptr = cpu_ptr
    _hip.stream = ops.stream = lambda: None
    _hip.stream_on = ops.stream_on = lambda device: None
    _hip.device_tensor = lambda t: t
    try:
        yield handle
    finally:
        handle.dream_conv3x3_winograd_set_max_workgroups(0)
        handle.dream_conv3x3_winograd4_set_max_workgroups(0)
        _hip._lib, _hip.ptr, _hip.stream, ops.ptr, ops.stream, _hip.device_tensor, _hip.stream_on, ops.stream_on = saved
