#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace [+ PMC]) into a small text/CSV report that can be
committed under profiles/.  Usage: python tools/prof_summary.py gpurun_out/prof/r01_results.db out_prefix"""
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
