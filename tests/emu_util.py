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

    _hip._lib = handle
    _hip.ptr = ops.ptr = cpu_ptr
    _hip.stream = ops.stream = lambda: None
    _hip.stream_on = ops.stream_on = lambda device: None
    _hip.device_tensor = lambda t: t
    try:
        yield handle
    finally:
        _hip._lib, _hip.ptr, _hip.stream, ops.ptr, ops.stream, _hip.device_tensor, _hip.stream_on, ops.stream_on = saved
