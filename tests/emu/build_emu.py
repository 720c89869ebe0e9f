"""TEST INFRASTRUCTURE: builds tests/emu/libdream_emu.so -- the product's .hip sources compiled as host
C++ against the SIMT emulator headers in this directory (ROCm's clang++ as a plain x86 compiler)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libdream_emu.so")


def build(force=False):
    srcs = [os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"),
            os.path.join(HERE, "dream_cdna4.h")]
    csrc = os.path.join(ROOT, "dream_amd", "csrc")
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".inc"))]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(s) <= os.path.getmtime(OUT) for s in srcs):
        return OUT
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        clang = "clang++"
    cmd = [clang, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-pthread",
           "-Wno-unused-value", "-Wno-psabi", "-Wno-gnu-anonymous-struct", "-Wno-vla-cxx-extension",
           "-I", HERE, "-I", csrc, os.path.join(HERE, "emu_runtime.cpp")]
    cmd += sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip"))   # each its own TU
    cmd += ["-ldl", "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
