#!/usr/bin/env python3
"""Full-size parity figure (GPU box): PSNR of the bf16 / fp16 fused kernel against the fp32 HIP path (itself pinned to
the reference at 1e-4) on COMPLETE frames of every bench workload, identical in-kernel noise.  One JSON line each.
    python tools/full_frame_psnr.py [workload ...]"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else -10 * math.log10(mse)


def main():
    dev = torch.device("cuda")
    for wl in sys.argv[1:] or list(bench.WORKLOADS):
        outs = {}
        for prec in ("fp32", "bf16", "fp16"):
            spec, net, rend, rays = bench.build(wl, prec, dev)
            rend.forced_seed = 4242
            t0 = time.perf_counter()
            with torch.no_grad():
                o = rend(net, rays)
            torch.cuda.synchronize()
            lvl = o.fine if spec["Kf"] > 0 else o.coarse
            outs[prec] = (lvl.rgb.cpu(), o.coarse.rgb.cpu(), time.perf_counter() - t0)
        print(json.dumps({"workload": wl, "rays": int(rays.shape[1]),
                          "psnr_final_bf16": round(psnr(outs["bf16"][0], outs["fp32"][0]), 1),
                          "psnr_final_fp16": round(psnr(outs["fp16"][0], outs["fp32"][0]), 1),
                          "psnr_coarse_bf16": round(psnr(outs["bf16"][1], outs["fp32"][1]), 1),
                          "psnr_coarse_fp16": round(psnr(outs["fp16"][1], outs["fp32"][1]), 1),
                          "fp32_path_s": round(outs["fp32"][2], 2)}), flush=True)


if __name__ == "__main__":
    main()
