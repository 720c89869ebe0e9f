#!/usr/bin/env python
"""Non-MFMA instructions between consecutive MFMAs of the innermost loop of a kernel (hipcc -S listing): the issue work a wavefront
has to fit into its partner's MFMA time.   python tools/isa_gaps.py file.s <kernel substring>"""
import re
import sys


def main(path, key):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if key in l and l.startswith("_Z") and ":" in l][0]
    end = [i for i, l in enumerate(lines) if i > start and ".amdhsa_kernel" in l][0]
    body = lines[start:end]
    depth2 = [i for i, l in enumerate(body) if "Depth=2" in l and l.startswith(".LBB")]
    lo = depth2[0]
    hi = max(i for i, l in enumerate(body) if "s_cbranch" in l and i > depth2[-1] and i < depth2[-1] + 1500)
    gaps, cur = [], []
    for l in body[lo:hi + 1]:
        t = l.strip().split()
        if not t or not re.match(r"^(v_|s_|ds_|buffer_|global_)", t[0]):
            continue
        if t[0].startswith("v_mfma"):
            gaps.append(cur)
            cur = []
        else:
            cur.append(t[0])
    out = []
    for g in gaps:
        valu = sum(1 for x in g if x.startswith("v_"))
        mem = sum(1 for x in g if x.startswith(("buffer_", "ds_", "global_")))
        out.append("%d/%d/%d" % (valu, mem, len(g) - valu - mem))
    print("gaps (VALU/mem/scalar) before each MFMA:")
    for i in range(0, len(out), 16):
        print("  " + " ".join("%8s" % o for o in out[i:i + 16]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
