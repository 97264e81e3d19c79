#!/usr/bin/env python3
"""Training-step timing of the differentiable path: renderer forward with the tape + loss +
backward, at the reference's training shape (conf/default_mv.conf + train/train.py: SB objects x ray_batch_size
rays, 64 coarse + 32 fine samples, 1-3 source views).  Prints one JSON line per configuration.
    python tools/bench_train.py [--sb 4] [--rays 128] [--views 1 2] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402
import hip_util as hu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sb", type=int, default=4)
    ap.add_argument("--rays", type=int, default=128)
    ap.add_argument("--views", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"])
    a = ap.parse_args()
    for ns in a.views:
        spec = gu._case(seed=5, d_hidden=512, lat=[(256, 8, 8)], image=(128, 128), focal=131.25, NS=ns, SB=a.sb,
                        N=a.rays, Kc=64, Kf=32, Kfd=16)
        rays_np, poses_np = gu.make_inputs(spec)
        net = hu.build_net(spec, poses_np).train()
        net.train_precision = a.precision
        maps = [torch.from_numpy(x).cuda().requires_grad_(True) for x in gu.make_latents(spec)]
        net.encoder.set_latents(maps)
        rend = hu.build_renderer(spec)
        rays = torch.from_numpy(rays_np).cuda()
        tgt = torch.rand(a.sb, a.rays, 3, device="cuda")
        opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)

        def step():
            opt.zero_grad(set_to_none=True)
            out = rend(net, rays, want_weights=True)
            loss = ((out.coarse.rgb - tgt) ** 2).mean() + ((out.fine.rgb - tgt) ** 2).mean()
            loss.backward()
            opt.step()

        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        n_rays = a.sb * a.rays
        pts = n_rays * (64 + 96)
        flop_fwd = pts * 2 * (ns * 1987584 + 1050624)
        print(json.dumps({"what": "train_step", "sb": a.sb, "rays_per_obj": a.rays, "views": ns, "samples": "64+32",
                          "ms_per_step": round(ms, 2), "rays_per_s": round(n_rays / ms * 1e3),
                          "tflops_fwd_bwd": round(3 * flop_fwd / ms / 1e9, 1), "dtype": {"fp32": "f32", "bf16": "bf16 products, f32 accumulate", "bf16x3": "bf16x3 split products (fp32-class), f32 accumulate"}[a.precision]}))


if __name__ == "__main__":
    main()
