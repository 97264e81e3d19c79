"""CPU emulation (no GPU): which layers of the fused kernel need fp16 for SURVEY 8(c)'s 50 dB?

The fused kernel rounds every layer's two operands — the weights (pack time) and the layer's input activations (the
relu / convert snapshot) — to its 16-bit format and accumulates in fp32.  bf16 (7 mantissa bits) renders the DTU shape at
44-45 dB, fp16 (10 bits) at 61 dB, and fp16 costs ~5 % more time on the power-limited launch (profiles/r04_fp16_ovfl.txt:
the clock follows the operands' mantissa bits).  This script evaluates the oracle restatement of the network with a
16-bit format PER LAYER GROUP (operands rounded with torch's own fp16 / bf16 casts, products exact in fp32, fp32
accumulation) on a sample of the DTU frame, and reports PSNR of the composited pixels against the fp32 evaluation — i.e.
how much of the network could run in bf16 before the render drops under 50 dB.

    python tools/dev/mixed_precision_emul.py [n_rays]
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
import golden_util as gu  # noqa: E402
from oracle import pixelnerf_oracle as orc  # noqa: E402


def rnd(x, fmt):
    if fmt == "fp32":
        return x
    return x.to(torch.float16 if fmt == "fp16" else torch.bfloat16).float()


def resnetfc_q(sd, zx, d_latent, NS, P, fmt, n_blocks=5, combine_layer=3):
    """fmt: dict layer-group -> format; groups: lin_in, lin_z, b0..b4 (both layers of a block), lin_out."""
    def lin(x, k, f):
        return torch.addmm(sd[k + ".bias"], rnd(x, f), rnd(sd[k + ".weight"], f).t())
    z, x = zx[:, :d_latent], zx[:, d_latent:]
    x = lin(x, "lin_in", fmt["lin_in"])
    for b in range(n_blocks):
        if b == combine_layer and NS > 1:
            x = x.reshape(-1, NS, P, x.shape[-1]).mean(dim=1).reshape(-1, x.shape[-1])
        if d_latent > 0 and b < combine_layer:
            x = x + lin(z, f"lin_z.{b}", fmt["lin_z"])
        f = fmt[f"b{b}"]
        net = lin(torch.relu(x), f"blocks.{b}.fc_0", f)
        x = x + lin(torch.relu(net), f"blocks.{b}.fc_1", f)
    return lin(torch.relu(x), "lin_out", fmt["lin_out"])


def render(sd, cam, lat, rays, z, NS, fmt, white, far_):
    w2c, focal, c = cam
    SB, B, K = 1, rays.shape[0], z.shape[1]
    xyz = (rays[:, None, :3] + z[..., None] * rays[:, None, 3:6]).reshape(1, -1, 3)
    vd = rays[:, None, 3:6].expand(-1, K, -1).reshape(1, -1, 3)
    P = xyz.shape[1]
    rep = lambda t: t.unsqueeze(1).expand(-1, NS, *t.shape[1:]).reshape(-1, *t.shape[1:])
    x = rep(xyz)
    x_rot = torch.matmul(w2c[:, None, :3, :3], x.unsqueeze(-1))[..., 0]
    x_cam = x_rot + w2c[:, None, :3, 3]
    zf = x_rot.reshape(-1, 3)
    v = torch.matmul(w2c[:, None, :3, :3], rep(vd.reshape(1, P, 3, 1))).reshape(-1, 3)
    zf = torch.cat((orc.positional_encoding(zf), v), dim=1)
    uv = -x_cam[:, :, :2] / x_cam[:, :, 2:]
    uv = uv * focal.unsqueeze(1) + c.unsqueeze(1)
    latv = orc.index_latent(uv, lat)
    L = latv.shape[1]
    zx = torch.cat((latv.transpose(1, 2).reshape(-1, L), zf), dim=-1)
    o = resnetfc_q(sd, zx, L, NS, P, fmt).reshape(B, K, 4)
    rgb, sigma = torch.sigmoid(o[..., :3]), torch.relu(o[..., 3])
    delta = torch.cat([z[:, 1:] - z[:, :-1], far_ - z[:, -1:]], -1)
    alpha = 1 - torch.exp(-delta * sigma)
    T = torch.cumprod(torch.cat([torch.ones(B, 1), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    out = (w[..., None] * rgb).sum(1)
    if white:
        out = out + 1 - w.sum(-1, keepdim=True)
    return out


def psnr(a, b):
    return -10 * math.log10(float(((a.double() - b.double()) ** 2).mean()))


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    torch.set_num_threads(8)
    for wl in ("dtu_3view_400x300_k128", "srn_chairs_1view_128x128_k128"):
        w = bench.WORKLOADS[wl]
        W_img, H_img = w["side"] if isinstance(w["side"], tuple) else (w["side"], w["side"])
        spec = dict(gu._BASE)
        spec.update(seed=100, d_hidden=512, lat=w["lat"], NS=w["NS"], SB=1, image=(W_img, H_img), focal=w["focal"], N=0, Kc=w["Kc"],
                    Kf=0, Kfd=0, lindisp=w["lindisp"], white_bkgd=w["white"], use_code_viewdirs=False, z_near=w["z"][0],
                    z_far=w["z"][1], radius=w["radius"])
        poses = np.stack([gu.pose_spherical(30.0 * v, -20.0, w["radius"]) for v in range(w["NS"])])[None]
        cam = orc.encode_cameras(torch.from_numpy(poses), spec["focal"], None, W_img, H_img)
        lat = [torch.from_numpy(x) for x in gu.make_latents(spec)]
        sd = {k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, "coarse").items()}
        tgt = gu.pose_spherical(75.0, -25.0, w["radius"])
        g = torch.Generator().manual_seed(0)
        pix = torch.randperm(W_img * H_img, generator=g)[:n_rays].numpy()
        rays = torch.from_numpy(gu.pinhole_rays(tgt, W_img, H_img, w["focal"], w["z"][0], w["z"][1], pix))
        K = w["Kc"]
        t = (torch.arange(K)[None] + torch.rand(n_rays, K, generator=g)) / K
        near, far = w["z"]
        z = 1.0 / ((1 - t) / near + t / far) if w["lindisp"] else near * (1 - t) + far * t
        groups = ["lin_in", "lin_z", "b0", "b1", "b2", "b3", "b4", "lin_out"]
        allf = lambda f: {gk: f for gk in groups}
        with torch.no_grad():
            ref = render(sd, cam, lat, rays, z, w["NS"], allf("fp32"), w["white"], far)
            print(f"== {wl}: {n_rays} rays x {K} samples, NS = {w['NS']}")
            for name, fmt in [("all bf16", allf("bf16")), ("all fp16", allf("fp16"))]:
                print(f"{name:58s} {psnr(render(sd, cam, lat, rays, z, w['NS'], fmt, w['white'], far), ref):6.1f} dB", flush=True)
            # one group in fp16, the rest bf16: which group's rounding matters
            for gk in groups:
                fmt = allf("bf16"); fmt[gk] = "fp16"
                print(f"bf16 with {gk:8s} in fp16{'':33s} {psnr(render(sd, cam, lat, rays, z, w['NS'], fmt, w['white'], far), ref):6.1f} dB", flush=True)
            # the candidates: per-view part (lin_in, lin_z, b0-b2) in bf16, the part behind the view reduction in fp16; and the reverse
            for name, f16 in [("per-view part bf16 | b3, b4, lin_out fp16", ["b3", "b4", "lin_out"]),
                              ("per-view part bf16 | b2, b3, b4, lin_out fp16", ["b2", "b3", "b4", "lin_out"]),
                              ("lin_in, lin_z, b0, b1, b2 fp16 | b3, b4, lin_out bf16", ["lin_in", "lin_z", "b0", "b1", "b2"]),
                              ("b0 bf16 | rest fp16", ["lin_in", "lin_z", "b1", "b2", "b3", "b4", "lin_out"]),
                              ("b0, b1 bf16 | rest fp16", ["lin_in", "lin_z", "b2", "b3", "b4", "lin_out"])]:
                fmt = allf("bf16")
                for gk in f16:
                    fmt[gk] = "fp16"
                print(f"{name:58s} {psnr(render(sd, cam, lat, rays, z, w['NS'], fmt, w['white'], far), ref):6.1f} dB", flush=True)


if __name__ == "__main__":
    main()
