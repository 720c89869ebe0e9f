#!/usr/bin/env python
"""Per-kernel hardware counters from several rocprofv3 --pmc passes of ONE command (each pass: `--pmc <counters> --kernel-trace`),
one line per kernel name (template instance) that matches the filter: launches, average duration, and the averages of every counter
collected, plus the derived fractions the round-6 question about the 1x1 GEMMs needs (is a launch MFMA-, wait-, L2- or fill-bound?):

  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines)
  wait, stall = SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES (parked at s_waitcnt / waiting to issue)
  occupancy   = SQ_WAVE_CYCLES / (4 x SQ_BUSY_CYCLES / 32 x 1024)  (average wavefronts per SIMD while the kernel runs; quad-cycles)
  l2_hit      = TCC_HIT / (TCC_HIT + TCC_MISS);  l1_miss_bytes = TCP_TCC_READ_REQ x 64 B (gfx950: 128-B lines counted per 64 B)
  l2_latency  = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ (cycles per L1 -> L2 read request)

    python tools/pmc_kernels.py <filter substring[,substring..]> <pass dir> [<pass dir> ...] > profiles/r06_pmc_gemm1x1_in_step.txt
"""
import glob
import os
import sqlite3
import sys


def main():
    keys = sys.argv[1].split(",")
    acc = {}
    dur = {}
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            cur = sqlite3.connect(f).cursor()
            try:
                rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                        "group by kernel_name, counter_name"))
            except sqlite3.Error as e:
                print("skip", f, e, file=sys.stderr)
                continue
            for name, counter, total, n in rows:
                if not any(k in name for k in keys):
                    continue
                k = acc.setdefault(name, {})
                k[counter] = (k.get(counter, (0.0, 0))[0] + float(total), k.get(counter, (0.0, 0))[1] + int(n))
            try:
                for name, n, tot in cur.execute("select name, count(*), sum(duration) from kernels group by name"):
                    if any(k in name for k in keys):
                        a = dur.setdefault(name, [0, 0.0])
                        a[0] += n
                        a[1] += tot
            except sqlite3.Error:
                pass
    for name in sorted(acc, key=lambda n: -dur.get(n, [0, 0.0])[1]):
        c = {k: v[0] / max(v[1], 1) for k, v in acc[name].items()}
        n = max(v[1] for v in acc[name].values())
        short = name.replace("void ", "").replace("(anonymous namespace)::", "")
        short = short[:short.index(">(") + 1] if ">(" in short else short.split("(")[0]
        d = dur.get(name, [0, 0.0])
        print("== %s   launches/pass %d   avg duration under PMC %.1f us" % (short, n, d[1] / max(d[0], 1) / 1e3))
        der = []
        busy = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
        if busy and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            der.append("mfma_util %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * busy)))
        if c.get("SQ_WAVE_CYCLES"):
            wc = c["SQ_WAVE_CYCLES"]
            for k, lab in (("SQ_WAIT_ANY", "wait"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_ACTIVE_INST_ANY", "active")):
                if k in c:
                    der.append("%s %.3f" % (lab, c[k] / wc))
            if busy:
                der.append("waves_per_simd %.2f" % (wc * 4.0 / (busy * 1024.0)))
        if busy:
            der.append("kernel_cycles %.0f (%.1f us at 2.4 GHz)" % (busy, busy / 2400.0))
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            der.append("l2_hit %.3f" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
            der.append("l2_miss_MB %.1f" % (c["TCC_MISS_sum"] * 128.0 / 1e6))
        if c.get("TCP_TCC_READ_REQ_sum"):
            der.append("l1_to_l2_read_MB %.1f" % (c["TCP_TCC_READ_REQ_sum"] * 64.0 / 1e6))
            if c.get("TCP_TCC_READ_REQ_LATENCY_sum"):
                der.append("l2_read_latency %.0f cyc" % (c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]))
        if c.get("SQ_INSTS_MFMA") and c.get("SQ_INSTS_VMEM_RD"):
            der.append("mfma_per_vmem_rd %.1f" % (c["SQ_INSTS_MFMA"] / c["SQ_INSTS_VMEM_RD"]))
        print("   " + "   ".join(der))
        print("   " + "  ".join("%s=%.4g" % (k, v) for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()
