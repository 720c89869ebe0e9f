#!/usr/bin/env python
"""Instruction mix of the innermost loop(s) of one kernel in a hipcc -S listing:  python tools/isa_loop_mix.py file.s <kernel substring>"""
import collections
import re
import sys


def main(path, key):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(":") or (key in l and l.startswith("_Z") and ":" in l)][0]
    end = [i for i, l in enumerate(lines) if i > start and ".amdhsa_kernel" in l][0]
    body = lines[start:end]
    depth2 = [i for i, l in enumerate(body) if "Depth=2" in l and l.startswith(".LBB")]
    lo = depth2[0]
    hi = max(i for i, l in enumerate(body) if "s_cbranch" in l and i > depth2[-1] and i < depth2[-1] + 1500)
    c = collections.Counter()
    for l in body[lo:hi + 1]:
        t = l.strip().split()
        if t and re.match(r"^(v_|s_|ds_|buffer_|global_|scratch_)", t[0]):
            c[t[0]] += 1
    print(path, "lines", lo, hi, "instructions", sum(c.values()))
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print("  %4d %s" % (v, k))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
