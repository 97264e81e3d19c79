"""Debug (GPU box): fused kernel vs fp32 path as a function of the number of tiles (workgroups)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_util as gu
from hip_util import build_net
name = sys.argv[1] if len(sys.argv) > 1 else "full_ns1"
spec = dict(gu.CASES[name]); fx = gu.load_fixture(name)
g = torch.Generator().manual_seed(2)
N = 128 * 600
xyz_all = ((torch.rand(1, N, 3, generator=g) - 0.5) * 1.6).cuda()
vd_all = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=g), dim=-1).cuda()
net32 = build_net(spec, fx["poses"], "cuda", "fp32")
nets = {p: build_net(spec, fx["poses"], "cuda", p) for p in ("fp16", "bf16")}
for nt in (1, 2, 3, 8, 32, 64, 128, 255, 256, 257, 512, 600):
    xyz, vd = xyz_all[:, :128 * nt].contiguous(), vd_all[:, :128 * nt].contiguous()
    ref = net32(xyz, viewdirs=vd).cpu().numpy()[0]
    line = f"{name} tiles {nt:4d}:"
    for p, net in nets.items():
        outs = [net(xyz, viewdirs=vd).cpu().numpy()[0] for _ in range(3)]
        err = np.abs(outs[0][:, :3] - ref[:, :3]).max(-1).reshape(-1, 128).mean(1)
        det = max(np.abs(outs[0] - o).max() for o in outs[1:])
        line += f"  {p}: bad tiles {(err > 0.01).sum():3d}/{nt} mean {err.mean():.5f} rep-to-rep {det:.1e}"
    print(line, flush=True)
