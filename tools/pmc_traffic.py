#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the default bench command with
--kernel-trace only) into profiles/<round>_pmc_traffic.json: HBM bytes per launch of the dominant kernel.
Usage: python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/*.db gpurun_out/pmc_WRITE_SIZE/*.db profiles/r01_pmc_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE counts 128-B requests as 64 B, so the raw
value is doubled; the max-pool kernel (pure streaming, algorithmic bytes known exactly) is reported beside it as the
calibration of that rule on this run."""
import hashlib
import json
import os
import sqlite3
import sys


def bench_py_sha():
    """bench.py reports roofline.traffic from this summary only when it was measured with the same bench.py."""
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]

B = 128
# vgg_q at 400x400: (cin, cout, H) per MFMA conv launch, pooled = 2x2 max-pool fused into the store
LAYERS = [(64, 64, 400, True), (64, 128, 200, False), (128, 128, 200, True), (128, 256, 100, False), (256, 256, 100, False),
          (256, 256, 100, False), (256, 256, 100, False), (256, 512, 50, False), (512, 512, 50, False), (512, 512, 50, False),
          (512, 512, 50, False), (512, 512, 25, False), (512, 512, 25, False), (512, 512, 25, False), (512, 512, 25, False),
          (512, 256, 50, "up"), (256, 256, 50, False), (256, 128, 100, "up"), (128, 64, 100, False), (64, 64, 100, False),
          (64, 32, 100, False), (32, 7, 100, False)]
# the two "up" layers (nearest x2 upsample + conv) run as four sub-pixel phase launches each: 28 kernel launches per pass
N_LAUNCHES = sum(4 if m == "up" else 1 for _, _, _, m in LAYERS)


def algorithmic_bytes():
    total = 0
    for cin, cout, h, mode in LAYERS:
        hin = h // 2 if mode == "up" else h              # fused nearest-x2 upsample reads the half-resolution tensor
        hout = h // 2 if mode is True else h
        total += 4 * B * (hin * hin * cin + hout * hout * cout) + 4 * 9 * cin * cout
    return total / N_LAUNCHES


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                       "group by kernel_name order by sum(value) desc", (counter,))
    return [{"kernel": r[0][:90], "dispatches": r[1], "sum_kib": r[2]} for r in rows]


def main():
    fetch_db, write_db, out = sys.argv[1:4]
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")

    def family(rows, key):
        sel = [r for r in rows if key in r["kernel"]]
        return sum(r["dispatches"] for r in sel), sum(r["sum_kib"] for r in sel) * 1024.0

    # the conv kernel families of the fp32 path: the Winograd kernel and the direct implicit-GEMM kernel
    nf, fb = [a + b + c for a, b, c in zip(family(fetch, "mfma_kernel"), family(fetch, "wino_kernel"), family(fetch, "wino4_kernel"))]
    nw, wb = [a + b + c for a, b, c in zip(family(write, "mfma_kernel"), family(write, "wino_kernel"), family(write, "wino4_kernel"))]
    n_wino = family(fetch, "wino_kernel")[0]
    n_wino4 = family(fetch, "wino4_kernel")[0]
    npool, pool_fetch = family(fetch, "maxpool2_kernel")
    # un-fused pools of vgg_q at B=128: 256ch@100x100 and 512ch@50x50 inputs, one of each per forward pass
    pool_alg = 4.0 * B * (100 * 100 * 256 + 50 * 50 * 512) * (npool / 2.0)
    if len(sys.argv) > 4:                                   # generic mode: just the per-family totals of another workload
        fams = sys.argv[4].split(",")
        out_d = {"bench_py_sha": bench_py_sha(), "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py " + " ".join(sys.argv[5:]),
                 "units": "GB per profiled run; FETCH_SIZE doubled per the gfx950 rule", "families": {}}
        for fam in fams:
            n1, b1 = family(fetch, fam)
            n2, b2 = family(write, fam)
            out_d["families"][fam] = {"dispatches": n1, "fetch_gb_corrected": 2.0 * b1 / 1e9, "write_gb": b2 / 1e9}
        out_d["raw"] = {"FETCH_SIZE": fetch[:16], "WRITE_SIZE": write[:16]}
        with open(out, "w") as f:
            json.dump(out_d, f, indent=1)
        print(json.dumps(out_d["families"], indent=1))
        return
    # second calibration, on a store stream: the first conv writes B x 400 x 400 x 64 floats per dispatch and nothing else
    nfirst, first_write = family(write, "conv3x3_first_kernel")
    first_alg = 4.0 * B * 400 * 400 * 64 * nfirst
    res = {
        "bench_py_sha": bench_py_sha(),
        "command": "DREAM_BENCH_PMC_CALIBRATE=1 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 1 "
                   "--no-cpu-baseline --no-split-leg --no-secondary (one PMC counter per run; the calibration launches are two "
                   "stand-alone max-pools outside the timed region)",
        "workload": "DREAM-vgg-Q inference B=128 400x400",
        "units": "counters are KiB; FETCH_SIZE doubled per the gfx950 rule (see calibration)",
        "calibration_maxpool2_kernel": {"dispatches": npool, "fetch_gb_raw": pool_fetch / 1e9,
                                        "fetch_gb_algorithmic": pool_alg / 1e9,
                                        "raw_to_algorithmic": pool_alg / pool_fetch if pool_fetch else None},
        "calibration_conv3x3_first_kernel_writes": {"dispatches": nfirst, "write_gb_raw": first_write / 1e9,
                                                    "write_gb_algorithmic": first_alg / 1e9,
                                                    "raw_to_algorithmic": first_alg / first_write if first_write else None},
        "conv_kernels": {
            "families": "conv_wino4_kernel (%d dispatches) + conv_wino_kernel (%d) + conv_mfma_kernel (%d)" % (n_wino4, n_wino, nf - n_wino - n_wino4),
            "dispatches": nf,
            "fetch_gb_per_launch_corrected": 2.0 * fb / nf / 1e9,
            "write_gb_per_launch": wb / nw / 1e9,
            "traffic_gb_per_launch": (2.0 * fb / nf + wb / nw) / 1e9,
            "algorithmic_gb_per_launch": algorithmic_bytes() / 1e9,
        },
        "raw": {"FETCH_SIZE": fetch[:12], "WRITE_SIZE": write[:12]},
    }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("calibration_maxpool2_kernel", "calibration_conv3x3_first_kernel_writes", "conv_kernels")}, indent=1))


if __name__ == "__main__":
    main()
