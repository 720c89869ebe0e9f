#!/usr/bin/env python
"""Is a kernel's machine code unchanged by an edit?  Compiles one .hip source of dream_amd/csrc at a git revision and in the working
tree for gfx950 and compares the assembly kernel by kernel (comments stripped, basic-block labels renumbered).  Runs without a GPU.

    python tools/isa_same.py HEAD~1 conv_wino.hip [substring of the mangled kernel names to report]

Why: kernels written against the edge of the register file (conv_wino_kernel<4,1,0>: 253-255 VGPRs) change their register
allocation when code is merely PRESENT in the translation unit -- round 4 added a BatchNorm-folding variant of the Winograd kernel
in the last hours, with no GPU time left to re-measure the plain kernels; this check showed all 15 of them instruction-identical to
the measured library (the body is included twice, conv_wino_body.inc, instead of being templated on the new feature)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dream_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I", os.path.join(CSRC, "include"), "-I", CSRC,
         "-I", os.path.join(ROOT, "include"), "-Wno-unused-result", "-x", "hip"]


def kernels(asm_path):
    txt = open(asm_path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
        body = re.sub(r";.*", "", m.group(2))
        body = re.sub(r"\.LBB\d+_", ".LBB_", body)
        body = re.sub(r"\.Ltmp\d+", ".Ltmp", body)
        out[m.group(1)] = (hashlib.md5(body.encode()).hexdigest()[:12], body.count("\n"))
    return out


def compile_to_asm(src_text_dir, name, tmp, tag):
    # the source must sit next to its headers / included bodies: compile a copy placed in csrc under a scratch name
    scratch = os.path.join(CSRC, "_isa_%s_%s" % (tag, name))
    with open(scratch, "w") as f:
        f.write(src_text_dir)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", scratch, "-save-temps=obj", "-o", os.path.join(tmp, tag + ".o")],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    finally:
        os.remove(scratch)
    stem = os.path.splitext(os.path.basename(scratch))[0]
    return os.path.join(tmp, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def main():
    rev, name = sys.argv[1], sys.argv[2]
    want = sys.argv[3] if len(sys.argv) > 3 else ""
    old = subprocess.check_output(["git", "-C", ROOT, "show", "%s:dream_amd/csrc/%s" % (rev, name)]).decode()
    new = open(os.path.join(CSRC, name)).read()
    with tempfile.TemporaryDirectory() as tmp:
        a = kernels(compile_to_asm(old, name, tmp, "old"))
        b = kernels(compile_to_asm(new, name, tmp, "new"))
    same = True
    for k in sorted(set(a) | set(b)):
        if want not in k:
            continue
        state = "only in %s" % ("old" if k in a else "new") if (k in a) != (k in b) else ("SAME" if a[k] == b[k] else "DIFFERENT")
        same &= state == "SAME" or state.startswith("only in new")
        print("%-11s %s" % (state, k))
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
