#!/usr/bin/env python
"""Which kernels of a hipcc -S listing contain load -> s_waitcnt vmcnt(0) -> store chains (a global load the compiler could not hoist above
the previous store because the two buffers may alias: one serialised memory round trip per element)?   python tools/isa_serial_loads.py file.s"""
import re
import sys


def main(path):
    lines = open(path).read().split("\n")
    name, seq, out = None, [], {}
    for l in lines:
        m = re.match(r"^(_Z\S+):", l)
        if m:
            if name:
                out[name] = seq
            name, seq = m.group(1), []
            continue
        t = l.strip().split()
        if not t or name is None:
            continue
        op = t[0]
        if op.startswith(("buffer_load", "global_load", "flat_load")):
            seq.append("L")
        elif op.startswith(("buffer_store", "global_store", "flat_store")):
            seq.append("S")
        elif op == "s_waitcnt" and "vmcnt" in l:
            seq.append("W" + l.split("vmcnt(")[1].split(")")[0])
    if name:
        out[name] = seq
    for k, seq in out.items():
        s = " ".join(seq)
        n = len(re.findall(r"S L W0 S", s))
        if n >= 3:
            print("%4d serialised load-store pairs  %s" % (n, k[:140]))


if __name__ == "__main__":
    main(sys.argv[1])
