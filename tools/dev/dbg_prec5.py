"""Debug (GPU box): which outputs differ between repeated fp16 launches."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_util as gu
from hip_util import build_net
name = "full_ns1"
spec = dict(gu.CASES[name]); fx = gu.load_fixture(name)
g = torch.Generator().manual_seed(2)
N = 128 * 64
xyz = ((torch.rand(1, N, 3, generator=g) - 0.5) * 1.6).cuda()
vd = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=g), dim=-1).cuda()
ref = build_net(spec, fx["poses"], "cuda", "fp32")(xyz, viewdirs=vd).cpu().numpy()[0]
net = build_net(spec, fx["poses"], "cuda", "fp16")
outs = [net(xyz, viewdirs=vd).cpu().numpy()[0] for _ in range(6)]
for r, o in enumerate(outs):
    d = np.abs(o - ref)
    bad = np.argwhere((d[:, :3].max(-1) > 0.02) | (d[:, 3] > 0.5 + 0.05 * np.abs(ref[:, 3])))[:, 0]
    print(f"rep {r}: {len(bad)} bad points:", [(int(i) // 128, (int(i) % 128) // 32, int(i) % 32) for i in bad[:12]], "(tile, wave, point)")
    for i in bad[:4]:
        print("    out", np.round(o[i], 3), "ref", np.round(ref[i], 3))
# characterise the susceptible points: projection into the 8x8 latent
import itertools
from oracle import pixelnerf_oracle as orc
cam = orc.encode_cameras(torch.from_numpy(fx["poses"]), spec["focal"], None, *spec["image"])
w2c, focal, c = cam
P = xyz.cpu()[0]
xr = P @ w2c[0, :3, :3].T
xc = xr + w2c[0, :3, 3]
uv = -xc[:, :2] / xc[:, 2:] * focal[0] + c[0]
allbad = set()
for o in outs:
    d = np.abs(o - ref)
    allbad |= set(np.argwhere((d[:, :3].max(-1) > 0.02) | (d[:, 3] > 0.5 + 0.05 * np.abs(ref[:, 3])))[:, 0].tolist())
inside = ((uv[:, 0] > 0) & (uv[:, 0] < 7) & (uv[:, 1] > 0) & (uv[:, 1] < 7)).numpy()
print("points with uv strictly inside the 8x8 map:", int(inside.sum()), "of", len(inside), "; bad ever:", len(allbad), "; bad & inside:", sum(inside[i] for i in allbad))
for i in sorted(allbad)[:12]:
    print("  pt", i, "uv", np.round(uv[i].numpy(), 2), "xyz", np.round(P[i].numpy(), 2), "|xr|max", float(xr[i].abs().max()))
ins_idx = np.argwhere(inside)[:, 0]
print("inside points never bad:", [int(i) for i in ins_idx if i not in allbad][:10])
