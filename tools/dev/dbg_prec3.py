"""Debug (GPU box): fused kernel vs fp32 path on many points; error by tile index (first tile of a workgroup vs later)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_util as gu
from hip_util import build_net
name = sys.argv[1] if len(sys.argv) > 1 else "full_ns1"
spec = dict(gu.CASES[name]); fx = gu.load_fixture(name)
g = torch.Generator().manual_seed(2)
n = 128 * 700
xyz = ((torch.rand(1, n, 3, generator=g) - 0.5) * 1.6).cuda()
vd = torch.nn.functional.normalize(torch.randn(1, n, 3, generator=g), dim=-1).cuda()
ref = build_net(spec, fx["poses"], "cuda", "fp32")(xyz, viewdirs=vd).cpu().numpy()[0]
for prec in ("fp16", "bf16"):
    net = build_net(spec, fx["poses"], "cuda", prec)
    for rep in range(2):
        out = net(xyz, viewdirs=vd).cpu().numpy()[0]
        err = np.abs(out[:, :3] - ref[:, :3]).max(-1).reshape(-1, 128)
        t = err.mean(1)
        print(f"{name} {prec} rep{rep}: tiles 0-255 mean err {t[:256].mean():.5f} (bad>0.01: {(t[:256] > 0.01).sum()}), tiles 256-511 {t[256:512].mean():.5f} ({(t[256:512] > 0.01).sum()}), tiles 512+ {t[512:].mean():.5f} ({(t[512:] > 0.01).sum()})  by pos/8 of bad tiles {np.round(err[t > 0.01].mean(0).reshape(16, 8).mean(1), 3) if (t > 0.01).any() else '-'}", flush=True)
