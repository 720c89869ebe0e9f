#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace [+ PMC]) into a small text/CSV report that can be
committed under profiles/.  Usage: python tools/prof_summary.py gpurun_out/prof/r01_results.db out_prefix [marker kernel] [marker
launches to skip]   (the last two restrict the concurrency summary to the timed steps)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    out = sys.argv[2]
    cur = db.cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                            "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
        for name, n, tot, avg, mn, mx in rows:
            f.write('"%s",%d,%.3f,%.1f,%.1f,%.1f,%.2f\n' % (name, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
        # template families (what bench.py's roofline.avg_launch_ms averages over)
        for fam in ("conv_wino4_kernel<", "conv_wino_kernel<", "conv_mfma_kernel<", "conv_f16x3_kernel<", "gemm1x1_kernel<", "wgrad1x1", "wgrad_kernel<", "wgrad_wino", "bn_"):
            sel = [r for r in rows if fam in r[0]]
            if sel:
                n = sum(r[1] for r in sel)
                tot = sum(r[2] for r in sel)
                f.write('"%s* (all of the family)",%d,%.3f,%.1f,%.1f,%.1f,%.2f\n' % (
                    fam, n, tot / 1e6, tot / n / 1e3, min(r[4] for r in sel) / 1e3, max(r[5] for r in sel) / 1e3, 100.0 * tot / total))
    print(open(out + "_kernel_stats.csv").read())
    # per-dispatch list of the dominant kernel family, with launch geometry
    with open(out + "_conv_dispatches.csv", "w") as f:
        f.write("kernel,start_ms,duration_us,grid_x,grid_y,lds_bytes,vgprs\n")
        t0 = None
        for name, start, dur, gx, gy, lds, vg in cur.execute(
                "select name, start, duration, grid_x, grid_y, lds_size, vgpr_count from kernels order by start"):
            t0 = start if t0 is None else t0
            if "conv_mfma" in name or "conv_wino" in name or "conv_f16x3" in name or "wgrad" in name:
                f.write('"%s",%.3f,%.1f,%d,%d,%d,%d\n' % (name.split("(")[0], (start - t0) / 1e6, dur / 1e3, gx, gy, lds, vg))
    # Concurrency: how much of the traced span has at least one kernel running (union of the dispatch intervals), how much has two or
    # more (main stream + weight-gradient stream), per-queue busy time -- "is the step bound by one dependent chain or by total work?"
    try:
        kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        qcol = "queue_id" if "queue_id" in kcols else ("stream_id" if "stream_id" in kcols else None)
        sel = "select start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")
        iv = [(r[0], r[1], r[2] if qcol else 0) for r in cur.execute(sel)]
        # window: the timed steps only -- from the end of the `skip`-th launch of the marker kernel (one per step: the optimizer for
        # training, the peak kernel for inference) to the end of its last launch; model construction and warm-up stay outside
        marker = sys.argv[3] if len(sys.argv) > 3 else None
        skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        note = "all dispatches (incl. set-up and warm-up)"
        if marker:
            marks = [r[0] for r in cur.execute("select end from kernels where name like ? order by start", ("%" + marker + "%",))]
            if len(marks) > skip + 1:
                lo, hi = marks[skip], marks[-1]
                iv = [(max(a, lo), min(e, hi), q) for a, e, q in iv if e > lo and a < hi]
                note = "%d steps: from the end of launch %d of %s to the end of its last launch" % (len(marks) - 1 - skip, skip + 1, marker)
        if iv:
            span = max(e for _, e, _ in iv) - min(a for a, _, _ in iv)
            ev = sorted([(a, 1) for a, _, _ in iv] + [(e, -1) for _, e, _ in iv])
            busy1 = busy2 = 0
            depth, last = 0, ev[0][0]
            for t, d in ev:
                if depth >= 1:
                    busy1 += t - last
                if depth >= 2:
                    busy2 += t - last
                depth += d
                last = t
            per_q = {}
            for a, e, q in iv:
                per_q[q] = per_q.get(q, 0) + (e - a)
            with open(out + "_concurrency.txt", "w") as f:
                f.write("window: %s\ntraced span %.2f ms\n" % (note, span / 1e6))
                f.write("sum of kernel durations %.2f ms = %.2f x the span\n" % (sum(e - a for a, e, _ in iv) / 1e6, sum(e - a for a, e, _ in iv) / span))
                f.write(">= 1 kernel running %.2f ms (%.1f %% of the span); >= 2 running %.2f ms (%.1f %%); GPU idle %.1f %%\n" % (
                    busy1 / 1e6, 100.0 * busy1 / span, busy2 / 1e6, 100.0 * busy2 / span, 100.0 * (1 - busy1 / span)))
                for q, b in sorted(per_q.items(), key=lambda kv: -kv[1]):
                    f.write("  %s %s: kernel time %.2f ms (%.1f %% of the span)\n" % (qcol or "queue", q, b / 1e6, 100.0 * b / span))
            print(open(out + "_concurrency.txt").read())
    except sqlite3.Error as e:
        print("no concurrency summary:", e)
    # PMC, if present
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        pm = list(cur.execute("select * from counters_collection limit 1"))
        if pm:
            print("counters_collection columns:", cols)
    except sqlite3.Error as e:
        print("no PMC data:", e)


if __name__ == "__main__":
    main()
