"""Parity sweep of the multi-view fused kernel (GPU box): fp16 kernel vs the fp32 HIP path over view counts, point
counts (incl. several tiles per workgroup and ragged tails), reductions, projected / general streams, map sizes."""
import sys, os, itertools
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch, golden_util as gu
from hip_util import build_net
def psnr(a, b):
    m = float(((a - b) ** 2).mean()); return 99.0 if m == 0 else -10 * np.log10(m)
worst = 99.0
g = torch.Generator().manual_seed(7)
for NS, n, comb, proj, lat in itertools.product((2, 3, 4), (1000, 33000, 70001), ("average", "max"), (True, False), ((256, 8, 8), (256, 12, 10))):
    spec = dict(gu.CASES["full_ns3"]); spec.update(NS=NS, combine_type=comb, lat=[lat], seed=50 + NS)
    poses = np.stack([gu.pose_spherical(25.0 * v, -20.0, spec["radius"]) for v in range(NS)])[None]
    xyz = ((torch.rand(1, n, 3, generator=g) - 0.5) * 1.6).cuda()
    vd = torch.nn.functional.normalize(torch.randn(1, n, 3, generator=g), dim=-1).cuda()
    ref = build_net(spec, poses, "cuda", "fp32")(xyz, viewdirs=vd).cpu().numpy()[..., :3]
    net = build_net(spec, poses, "cuda", "fp16"); net.project_latent = proj
    out = net(xyz, viewdirs=vd).cpu().numpy()[..., :3]
    p = psnr(out, ref); worst = min(worst, p)
    flag = "" if p >= 60.0 else "   <<<<<< LOW"
    print(f"NS={NS} n={n:6d} {comb:7s} proj={int(proj)} lat={lat[1]}x{lat[2]}: {p:5.1f} dB{flag}", flush=True)
print("worst", round(worst, 1))
