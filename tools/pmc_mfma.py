#!/usr/bin/env python
"""Hardware MFMA utilisation of the conv kernels from rocprofv3 --pmc passes (SQ counters; tools/gpu_r02_a.sh).

    python tools/pmc_mfma.py <pass1 dir> [<pass2 dir> ...]  > profiles/rNN_pmc_mfma.json

Per kernel family (conv_mfma_kernel, conv_f16x3_kernel, conv_wino_kernel, wgrad_kernel ...), summed over the dispatches of
the profiled run:
  mfma_util        = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32
                     shader engines (the SQ counters are summed over the 32 SEs; MI355X_MICROARCH.md: MFMA_BUSY counts
                     cycles, = 64 per v_mfma_f32_32x32x2_f32 and 32 per v_mfma_f32_32x32x16_f16 on its SIMD)
  mfma_insts       = SQ_INSTS_MFMA, cycles_per_mfma = MFMA_BUSY / INSTS_MFMA (64 / 32 expected)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt / s_barrier),
  issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waiting to issue: for an MFMA-bound kernel, on the matrix pipe)
"""
import glob
import json
import os
import re
import sqlite3
import sys

N_SE, N_SIMD = 32, 4 * 256


def family(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def main(dirs):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            cur = sqlite3.connect(f).cursor()
            try:
                rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                   "group by kernel_name, counter_name")
            except sqlite3.Error as e:
                print("skip", f, e, file=sys.stderr)
                continue
            for name, counter, total, n in rows:
                fam = acc.setdefault(family(name), {})
                fam[counter] = fam.get(counter, 0.0) + float(total)
                fam["_n_" + counter] = fam.get("_n_" + counter, 0) + int(n)
    out = {}
    for fam, c in sorted(acc.items()):
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) == 0:
            continue
        cyc = c["SQ_BUSY_CYCLES"] / N_SE
        row = {"dispatches": c["_n_SQ_BUSY_CYCLES"], "kernel_cycles_sum": cyc,
               "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"],
               "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc)}
        if c.get("SQ_INSTS_MFMA"):
            # the two passes run the same command: instruction counts of pass 2 belong to the same dispatches
            scale = c["_n_SQ_BUSY_CYCLES"] / max(c.get("_n_SQ_INSTS_MFMA", 1), 1)
            row["mfma_insts"] = c["SQ_INSTS_MFMA"] * scale
            row["cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_INSTS_MFMA"] * scale)
            row["valu_insts_per_mfma"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"]
            row["lds_insts_per_mfma"] = c.get("SQ_INSTS_LDS", 0.0) / c["SQ_INSTS_MFMA"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
            row["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
        if c.get("SQ_WAVE_CYCLES"):
            row["wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            row["issue_stall_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if c.get("GRBM_GUI_ACTIVE"):
            row["grbm_gui_active_sum"] = c["GRBM_GUI_ACTIVE"]
        out[fam] = row
    json.dump({"source": [os.path.basename(os.path.normpath(d)) for d in dirs], "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
