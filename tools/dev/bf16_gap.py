#!/usr/bin/env python3
"""Where does the low-precision kernel lose PSNR on a workload?  (GPU box.)  For a strided ray sample of each
workload: point-level error of rgb and sigma vs the fp32 HIP path on the SAME sample positions, the pixel-level PSNR,
and the pixel PSNR with only sigma / only rgb taken from the low-precision kernel (attribution through the
compositing stage).  Usage: python tools/dev/bf16_gap.py [workload ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else -10 * np.log10(mse)


def main():
    dev = torch.device("cuda", 0)
    names = sys.argv[1:] or ["srn_chairs_1view_128x128_k128", "dtu_3view_400x300_k128", "nmr_3view_64x64_k64+32"]
    for wl in names:
        spec, net32, rend32, rays = bench.build(wl, "fp32", dev)
        idx = torch.linspace(0, rays.shape[1] - 1, 4096).long().to(dev)
        sub = rays[:, idx].contiguous()
        rend32.forced_seed = 77
        rend32.keep_samples = True
        o32 = rend32(net32, sub, want_weights=True)
        z = o32.coarse.z.reshape(-1, spec["Kc"])
        r = sub.reshape(-1, 8)
        xyz = (r[:, None, :3] + z[..., None] * r[:, None, 3:6]).reshape(1, -1, 3)
        vd = r[:, None, 3:6].expand(-1, spec["Kc"], -1).reshape(1, -1, 3).contiguous()
        p32 = net32(xyz, coarse=True, viewdirs=vd).reshape(-1, spec["Kc"], 4)
        print(f"== {wl}: |xyz| max {float(xyz.abs().max()):.2f}  sigma mean {float(p32[..., 3].mean()):.2f} max {float(p32[..., 3].max()):.1f}  "
              f"delta median {float((z[:, 1:] - z[:, :-1]).median()):.4f}")
        for prec in ("bf16", "fp16"):
            _, net, rend, _ = bench.build(wl, prec, dev)
            for proj in ((True, False) if prec == "bf16" else (True,)):
                net.project_latent = proj
                p = net(xyz, coarse=True, viewdirs=vd).reshape(-1, spec["Kc"], 4)
                ds = (p[..., 3] - p32[..., 3])
                rend.forced_seed = 77
                o = rend(net, sub)
                line = (f"   {prec} proj={int(proj)}: points rgb {psnr(p[..., :3], p32[..., :3]):.1f} dB  sigma rms err {float(ds.pow(2).mean().sqrt()):.4f} "
                        f"(rel {float((ds.abs() / (1 + p32[..., 3].abs())).mean()):.5f})  pixels {psnr(o.coarse.rgb, o32.coarse.rgb):.1f} dB")
                # attribution: composite [rgb from A, sigma from B]
                mix_s = torch.cat([p32[..., :3], p[..., 3:]], -1).contiguous()
                mix_c = torch.cat([p[..., :3], p32[..., 3:]], -1).contiguous()
                _, rgb_s, _ = rend32._composite_native(r, z.contiguous(), mix_s)
                _, rgb_c, _ = rend32._composite_native(r, z.contiguous(), mix_c)
                _, rgb_0, _ = rend32._composite_native(r, z.contiguous(), p32.contiguous())
                line += f"  | only sigma low-prec {psnr(rgb_s, rgb_0):.1f} dB, only rgb low-prec {psnr(rgb_c, rgb_0):.1f} dB"
                print(line, flush=True)


if __name__ == "__main__":
    main()
