"""The C ABI without Python: tests/abi_c/abi_client.c (C99, gcc) includes include/dream_hip.h, links libdream_hip.so and
drives it with hipMalloc'ed buffers.  CPU: the header compiles as C and the client links and runs its no-device mode.
GPU: the client reproduces the reference's keypoints bit for bit on golden belief maps."""
import os
import subprocess

import numpy as np
import pytest

import cases
from dream_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi_c", "abi_client.c")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    _hip.lib()                                                      # raises if libdream_hip.so is not built
    exe = str(tmp_path_factory.mktemp("abi_c") / "abi_client")
    libdir = os.path.dirname(_hip.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
           SRC, "-o", exe, "-L", libdir, "-ldream_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_plain_c_and_client_links(client):
    out = subprocess.run([client, "symbols"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("abi ") and "11 selectable conv variants" in out and "= 256" in out


@pytest.mark.gpu
def test_c_client_reproduces_reference_keypoints(client, tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "peaks_golden.npz"))
    for name, (maps, off) in cases.peak_cases().items():
        maps = np.ascontiguousarray(maps, np.float32)
        want = np.ascontiguousarray(g[name + "/keypoints"].reshape(-1, 2), np.float32)
        path = str(tmp_path / (name + ".bin"))
        with open(path, "wb") as f:
            f.write(np.array(maps.shape, np.int32).tobytes())
            f.write(maps.tobytes())
            f.write(want.tobytes())
        r = subprocess.run([client, "peaks", path, repr(float(off))], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
