#!/usr/bin/env python
"""Is ops.winograd_tile's rule (F(4x4) only where the batch gives every resident workgroup two tile blocks) right?  The same network
timed with the rule (tile 0), with F(4x4) forced wherever the kernel applies (4) and with F(2x2) everywhere (2), round-robin.
python tools/tile_rule_check.py [arch mode batch]..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402
import bench  # noqa: E402
from dream_amd import ops  # noqa: E402


def main():
    specs = [a.split(":") for a in sys.argv[1:]] or [["vgg_f", "inference", "32"], ["vgg_q", "inference", "32"], ["vgg_q", "inference", "8"],
                                                      ["resnet_f", "inference", "32"], ["resnet_h", "train", "16"], ["vgg_q", "train", "16"]]
    ctx = bench.Context()
    ctx.single, ctx.single_ids, ctx.device_index, ctx.rank, ctx.world = False, [], 0, 0, 1
    for arch, mode, batch in specs:
        spec = dict(arch=arch, mode=mode, batch=int(batch), res=400, conv_algorithm="winograd")
        net, x, tgt, _ = bench.build_network(ctx, spec)

        def step():
            if mode == "train":
                return net.train([x], tgt)
            with torch.no_grad():
                return net.inference(x)
        best = {}
        for rnd in range(3):
            for tile in (0, 4, 2):
                ops.set_winograd_tile(tile)
                for _ in range(3 if rnd == 0 else 1):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                best[tile] = min(best.get(tile, 1e9), (time.perf_counter() - t0) / 5)
        ops.set_winograd_tile(0)
        print("%-9s %-9s b=%-3s  rule %.2f ms | F(4x4) forced %.2f ms | F(2x2) %.2f ms" % (arch, mode, batch, best[0] * 1e3, best[4] * 1e3, best[2] * 1e3), flush=True)
        del net, x, tgt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
