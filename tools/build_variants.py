#!/usr/bin/env python
"""A/B libraries beside the product one: build/lib<NAME>.so (tools/gpu_ab_libs.sh copies them over dream_amd/libdream_hip.so in turn).
    python tools/build_variants.py NAME[:scalar=a.hip,b.hip][:D=src.hip,-DX=1,-DY=2] ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main(specs):
    for spec in specs:
        parts = spec.split(":")
        name, scalar, defines = parts[0], set(), {}
        for part in parts[1:]:
            k, v = part.split("=", 1)
            if k == "scalar":
                scalar = set(g.HIP_SOURCES) if v == "all" else set(v.split(","))
            elif k == "D":
                items = v.split(",")
                defines[items[0]] = items[1:]
        lib = os.path.join(ROOT, "build", "lib%s.so" % name)
        g.build_hip(lib=lib, objdir=os.path.join(ROOT, "build", "obj_" + name), scalar_f32=scalar, defines=defines, verbose=False)
        print("built", lib, "scalar:", sorted(scalar), "defines:", defines)


if __name__ == "__main__":
    main(sys.argv[1:])
