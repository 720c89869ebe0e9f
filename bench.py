#!/usr/bin/env python
"""Benchmark of the DREAM belief-map hot path on MI355X (driver contract: see the task brief).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): DREAM-vgg-Q inference, batch 128 of synthetic 400x400 frames
already resident in HBM, one "step" = DreamNetwork.inference(x) = CNN forward + peak extraction,
result = [B,K,2] float32 keypoints on the host (the reference's return contract).  With N ranks each
rank processes its own batch of 128 (embarrassingly parallel, no data-path collective): weak scaling.

One JSON line on rank 0.  `roofline` is measured live: every launch of the dominant kernel
(conv_mfma_kernel, 22 launches per step for vgg_q) is bracketed by HIP events on the launch stream inside
the timed region; achieved = algorithmic FLOPs of those launches / their summed duration.
`cpu_baseline` times the CPU oracle (torch-CPU restatement of the reference + NumPy peak path) on a
bounded sample of the same workload on this host's cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak (not the 2:1-sparse figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU (weak scaling, the default)")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: this many frames in total, split evenly over "
                    "the GPUs (BASELINE configs[3]: 128 over 8, configs[4]: 256 over 8); overrides --batch")
    ap.add_argument("--single-process", action="store_true", help="N GPUs from ONE process through training.platform.gpu_ids "
                    "(the reference's nn.DataParallel contract, dream_amd/data_parallel.py) instead of one process per GPU")
    ap.add_argument("--conv-algorithm", choices=["winograd", "direct"], default="winograd",
                    help="fp32 3x3 stride-1 convs: Winograd F(2x2,3x3) on the fp32 MFMA (default) or the direct implicit GEMM")
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--mode", choices=["inference", "train"], default="inference")
    ap.add_argument("--precision", choices=["fp32", "fp16x3"], default="fp32",
                    help="conv kernel for inference: exact fp32 MFMA, or split-precision fp16x3 (fp32 in/out, fp32-class error)")
    ap.add_argument("--arch", choices=["vgg_q", "vgg_f", "resnet_h", "resnet_f"], default="vgg_q")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="inference through DreamNetwork.hip_graph (hipGraph replay; "
                    "per-launch HIP events are not recorded then)")
    ap.add_argument("--no-split-leg", action="store_true", help="skip the informational fp16x3 leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs of this same command, FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md and calibrated on the max-pool kernel).  Counters cannot be read from inside the
    process, so the latest committed profile summary is reported; null for any other workload."""
    for name in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return None
    if args.arch != "vgg_q" or args.mode != "inference" or args.batch != 128 or args.res != 400:
        return None
    with open(path) as f:
        d = json.load(f)
    key = "conv_kernels" if "conv_kernels" in d else "conv_mfma_kernel"
    if args.conv_algorithm == "direct" and key == "conv_kernels":
        return None
    return d[key]["traffic_gb_per_launch"] * 1e9


ARCH_K = {"vgg_q": (7, "panda"), "vgg_f": (7, "panda"), "resnet_h": (7, "panda"), "resnet_f": (17, "baxter")}


def synthetic_weights(state_dict):
    """Random-init weights of the architecture (there are no checkpoints here): fan-in scaled uniform conv / deconv
    weights so activations stay O(1) through the 23-100 layers and the belief maps cross the 0.01 peak threshold,
    non-trivial BatchNorm statistics so the folded eval-mode BN is exercised.  Deterministic per tensor name."""
    import zlib
    import torch
    out = {}
    for key, t in state_dict.items():
        g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)

        def uni(lo, hi):
            return torch.rand(tuple(t.shape), generator=g, dtype=torch.float64) * (hi - lo) + lo
        if key.endswith("num_batches_tracked"):
            v = torch.zeros_like(t)
        elif key.endswith("running_mean"):
            v = uni(-0.1, 0.1)
        elif key.endswith("running_var"):
            v = uni(0.8, 1.2)
        elif t.dim() == 4:
            transposed = "deconv" in key or ("upsample" in key and t.shape[-1] == 4)      # ConvTranspose2d: [Cin,Cout,k,k]
            fan_in = t.shape[0] * t.shape[2] * t.shape[3] / 4.0 if transposed else t.shape[1] * t.shape[2] * t.shape[3]
            b = (6.0 / fan_in) ** 0.5
            v = uni(-b, b)
        elif key.endswith("beta"):
            v = t.detach().cpu()
        elif key.endswith("weight"):                                                      # BatchNorm gamma
            v = uni(0.8, 1.2) * (0.25 if ("bn3." in key or "downsample.1." in key) else 1.0)
        else:
            v = uni(-0.05, 0.05)
        out[key] = v.to(t.dtype)
    return out


def cpu_baseline(arch, res, seconds):
    """CPU oracle on a bounded sample of the same workload: batches of 4 frames, forward + peaks."""
    import torch
    import cases
    from oracle import models as omodels, peaks as opeaks
    model = omodels.build_model(arch, ARCH_K[arch][0])
    model.load_state_dict(omodels.recipe_weights(model.state_dict()))
    model.eval()
    bs = 4
    x = torch.from_numpy(cases.image_batch(bs, res, res, seed=0))

    def one():
        with torch.no_grad():
            maps = model(x)[0].numpy()
        return opeaks.keypoints_from_belief_maps(maps, 0.0 if maps.shape[-1] >= 400 else 0.4395)

    one()                                   # warm-up
    t0 = time.time()
    n = 0
    while True:
        one()
        n += bs
        if time.time() - t0 >= seconds or n >= 64:
            break
    dt = time.time() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d frames of %dx%d (batches of %d), oracle torch-CPU %s forward + NumPy peak "
                      "extraction, %.1f s" % (n, res, res, bs, arch, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import cases
    import dream_amd
    from dream_amd import ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    single = args.single_process and world == 1 and args.gpus > 1
    n_dev = args.gpus if single else world
    if args.global_batch:
        assert args.global_batch % max(n_dev, 1) == 0, "--global-batch must divide evenly over the GPUs"
        args.batch = args.global_batch // max(n_dev, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # One process per GPU over RCCL.  (Rehearsal of the N > 1 path on a single-GPU box: DREAM_BENCH_BACKEND=gloo lets
    # the ranks share device 0 -- RCCL refuses two ranks on one device; numbers from such a run mean nothing.)
    backend = os.environ.get("DREAM_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()   # == local_rank unless the launcher exposes one device per rank
    torch.cuda.set_device(device_index)
    if world == 1 and os.environ.get("DREAM_FORCE_REDUCER"):      # rehearsal: exercise the gradient exchange with one rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group("gloo", rank=0, world_size=1)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node == --gpus"
    total_batch = args.batch * (n_dev if single else 1)            # frames this PROCESS handles per step

    n_kp, manip = ARCH_K[args.arch]
    cfg = dream_amd.default_network_config(args.arch, manip, batch_size=args.batch)
    cfg["training"]["config"]["net_input_resolution"] = [args.res, args.res]
    if single:                                      # DREAM_BENCH_GPU_IDS=0,0: rehearsal of the N-replica path on a one-GPU box
        ids = os.environ.get("DREAM_BENCH_GPU_IDS")
        cfg["training"]["platform"]["gpu_ids"] = [int(v) for v in ids.split(",")] if ids else list(range(args.gpus))
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        net = dream_amd.create_network_from_config_data(cfg)
    net.model.load_state_dict(synthetic_weights(net.model.state_dict()))
    net.model.module.conv_algorithm = args.conv_algorithm

    x = torch.from_numpy(cases.image_batch(total_batch, args.res, args.res, seed=rank)).cuda()
    if args.mode == "train":
        net.enable_training()
        ow, oh = net.trained_net_output_resolution()
        tgt = torch.from_numpy(cases.target_batch(total_batch, n_kp, (ow, oh), in_wh=(args.res, args.res), seed=rank)).cuda()
    else:
        net.enable_evaluation()
        net.hip_graph = bool(args.graph)
        if args.precision != "fp32":
            net.model.module.precision = args.precision

    # ---- per-launch timing of the dominant kernel (HIP events on the launch stream) ------------------
    conv_events = []          # (start, end, flops)
    recording = [False]

    def timed(orig, flops_of, kernel_launches=1, executed=1.0):
        """executed: executed MACs / direct-algorithm MACs of this operator (Winograd: 16 / 36), or a callable of the kwargs."""
        def wrapper(*a, **k):
            if not recording[0] or k.get("relu_mask") is not None:     # conv3x3(relu_mask=..) forwards to conv2d: timed there
                return orig(*a, **k)
            s_ev = torch.cuda.Event(enable_timing=True)
            e_ev = torch.cuda.Event(enable_timing=True)
            s_ev.record()
            y = orig(*a, **k)
            e_ev.record()
            fl = flops_of(y, *a, **k)
            conv_events.append((s_ev, e_ev, fl, kernel_launches, fl * (executed(k) if callable(executed) else executed)))
            return y
        return wrapper

    # algorithmic FLOPs of one launch = 2 * outputs * (input channels * taps); y is NHWC or NCHW [B,...]
    def pooled(flags):                    # a fused 2x2 max-pool stores 1/4 of the conv outputs it computed
        return 4.0 if flags & ops.CONV_POOL2 else 1.0

    ops.conv3x3 = timed(ops.conv3x3, lambda y, x, packed, bias, cout, flags=0, relu_mask=None: 2.0 * y.numel() * pooled(flags) * x.shape[3] * 9)
    ops.conv2d = timed(ops.conv2d, lambda y, x, packed, cout, ksize, stride=1, scale=None, shift=None, residual=None, flags=0:
                       2.0 * y.numel() * pooled(flags) * x.shape[3] * ksize * ksize)
    ops.conv3x3_winograd = timed(ops.conv3x3_winograd, lambda y, x, u, cout, scale=None, shift=None, residual=None, flags=0:
                                 2.0 * y.numel() * pooled(flags) * x.shape[3] * 9, executed=16.0 / 36.0)
    ops.conv1x1 = timed(ops.conv1x1, lambda y, x, packed, cout, *a, **k: 2.0 * y.numel() * x.shape[3])
    ops.conv_transpose3x3s2 = timed(ops.conv_transpose3x3s2, lambda y, x, packed, bias, cout, *a, **k: 2.0 * x.numel() * cout * 9,
                                    kernel_launches=4)        # the sub-pixel ops are four kernel launches each
    ops.conv_transpose4x4s2 = timed(ops.conv_transpose4x4s2, lambda y, x, packed, cout, *a, **k: 2.0 * x.numel() * cout * k.get("direct_taps", 16),
                                    kernel_launches=4, executed=lambda k: 16.0 / k.get("direct_taps", 16))
    ops.conv_transpose4x4s2_winograd = timed(ops.conv_transpose4x4s2_winograd, lambda y, x, u4, cout, *a, **k: 2.0 * x.numel() * cout * k.get("direct_taps", 16),
                                             kernel_launches=4, executed=lambda k: 9.0 / k.get("direct_taps", 16))     # minimal filtering: 9 multiplications
    ops.conv4x4s2_winograd = timed(ops.conv4x4s2_winograd, lambda y, dy, u4, cin: 2.0 * y.numel() * dy.shape[3] * 16,
                                   kernel_launches=4, executed=9.0 / 16.0)
    ops.conv_transpose3x3s2_f16x3 = timed(ops.conv_transpose3x3s2_f16x3, lambda y, x, amax, p16, cout, *a, **k: 2.0 * x.numel() * cout * 9,
                                          kernel_launches=4)
    ops.conv_transpose4x4s2_f16x3 = timed(ops.conv_transpose4x4s2_f16x3, lambda y, x, amax, p16, cout, *a, **k: 2.0 * x.numel() * cout * k.get("direct_taps", 16),
                                          kernel_launches=4, executed=lambda k: 16.0 / k.get("direct_taps", 16))
    ops.conv2d_amax = timed(ops.conv2d_amax, lambda y, x, packed, cout, ksize, *a, **k: 2.0 * y[0].numel() * x.shape[3] * ksize * ksize)
    ops.conv2d_f16x3 = timed(ops.conv2d_f16x3,
                             lambda y, x, amax, p16, cout, ksize, scale=None, shift=None, residual=None, flags=0, want_amax=True:
                             2.0 * y[0].numel() * pooled(flags) * x.shape[3] * ksize * ksize)

    def step():
        if args.mode == "train":
            return net.train([x], tgt)
        with torch.no_grad():
            return net.inference(x)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_region():
        """W warm-up steps, then exactly K steps between barrier+synchronize pairs; max over ranks."""
        del conv_events[:]
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        barrier()
        recording[0] = not args.graph          # events cannot be timed inside a captured graph
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        recording[0] = False
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        ms = sum(ev[0].elapsed_time(ev[1]) for ev in conv_events)
        fl = sum(ev[2] for ev in conv_events)
        executed_flops[0] = sum(ev[4] for ev in conv_events)
        return dt, ms, fl, sum(ev[3] for ev in conv_events), out

    executed_flops = [0.0]
    dt, conv_ms, conv_flops, n_launch, out_main = timed_region()
    conv_executed = executed_flops[0]
    # roofline peak: the fp32 MFMA rate for the exact kernel; for the split kernel every algorithmic MAC costs three
    # fp16 MFMA MACs, so its ceiling in ALGORITHMIC flops is the dense fp16 MFMA peak / 3
    peak = PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" or args.mode == "train" else PEAK_F16_MFMA_TFLOPS / 3.0

    # second, informational leg: the same workload on the split-precision conv kernel (fp32 in/out, 3 fp16 MFMAs per
    # product).  The headline `value` stays the exact-fp32 path unless --precision fp16x3 is given explicitly.
    split = None
    if args.mode == "inference" and args.precision == "fp32" and not args.no_split_leg:
        net.model.module.precision = "fp16x3"
        dt2, ms2, fl2, n2, out2 = timed_region()
        net.model.module.precision = "fp32"
        diff = float((out_main[0] - out2[0]).abs().max())
        scale = max(1.0, float(out_main[0].abs().max()))
        k32, k16 = out_main[1], out2[1]
        both = (k32 != -999.999) & (k16 != -999.999)
        split = {"value": total_batch * args.steps * world / dt2, "unit": "frames/s", "ms_per_step": dt2 / args.steps * 1e3,
                 "dtype": "f32 in/out; products as 3 x f16 MFMA (hi*hi + hi*lo + lo*hi), f32 accumulate",
                 "roofline": {"bound": "mfma", "kernel": "conv_f16x3_kernel",
                              "achieved": fl2 / (ms2 * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3.0, "unit": "TFLOP/s",
                              "frac": fl2 / (ms2 * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0), "launches": n2,
                              "note": "algorithmic FLOPs; each costs 3 f16 MFMA MACs, so the ceiling is 2500/3 TFLOP/s"},
                 "max_abs_diff_vs_fp32_path": diff, "tolerance": 1e-4 * scale,
                 "max_keypoint_diff_px": float((k32 - k16).abs()[both].max()) if bool(both.any()) else 0.0,
                 "detections_agree": float(((k32 == -999.999) == (k16 == -999.999)).float().mean())}

    if rank == 0:
        frames = total_batch * args.steps * world
        line = {
            "metric": "frames/s DREAM-%s %dx%d b=%d %s" % (args.arch.replace("_", "-"), args.res, args.res, args.batch, args.mode),
            "value": frames / dt, "unit": "frames/s", "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f32 (f16x3 split MFMA, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "DREAM-%s (%s, %d keypoints) %s, batch %d per GPU, synthetic %dx%d RGB frames "
                                   "resident in HBM; CNN forward + belief-map peak extraction%s"
                                   % (args.arch, manip, n_kp, args.mode, args.batch, args.res, args.res,
                                      " (BASELINE.json configs[%d])" % (2 if args.mode == "train" else 1)
                                      if args.arch == "vgg_q" else ""),
                       "batch_per_gpu": args.batch, "resolution": [args.res, args.res],
                       "parallelism": "dp%d%s" % (n_dev, " (single process, gpu_ids)" if single else ""),
                       "conv_algorithm": (("winograd F(2x2,3x3) for the stride-1 3x3 convs with >= 64 output channels, direct "
                                           "implicit GEMM elsewhere" if args.conv_algorithm == "winograd" else "direct implicit GEMM")
                                          + ("; stride-1 1x1 convs as LDS-free GEMMs" if args.arch.startswith("resnet") else ""))},
            "roofline": {
                "bound": "mfma",
                "kernel": (("conv_wino_kernel + conv_mfma_kernel" if args.conv_algorithm == "winograd" else "conv_mfma_kernel")
                           + (" + gemm1x1_kernel" if args.arch.startswith("resnet") else ""))
                if args.precision == "fp32" or args.mode == "train" else "conv_f16x3_kernel",
                "achieved": conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None,
                "peak": peak, "unit": "TFLOP/s",
                "frac": (conv_flops / (conv_ms * 1e-3) / 1e12 / peak) if conv_ms > 0 else None,
                # frac above counts the DIRECT algorithm's FLOPs; executed_frac counts the multiplications the
                # kernels actually issue (Winograd 16/36 of direct, upsample+conv as a 4x4 transposed conv 16/36)
                "executed_frac": (conv_executed / (conv_ms * 1e-3) / 1e12 / peak) if conv_ms > 0 else None,
                "executed_over_direct": conv_executed / conv_flops if conv_flops > 0 else None,
                "traffic": pmc_traffic(args), "traffic_unit": "bytes/launch (PMC, profiles/r0N_pmc_traffic.json)",
                "launches": n_launch, "avg_launch_ms": conv_ms / max(n_launch, 1),
                "algorithmic_gflop_per_launch": conv_flops / max(n_launch, 1) / 1e9,
                "share_of_step_time": conv_ms * 1e-3 / dt,
            },
        }
        if split is not None:
            line["split_precision"] = split
        if world == 1 and not args.no_cpu_baseline and args.mode == "inference":
            line["cpu_baseline"] = cpu_baseline(args.arch, args.res, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
