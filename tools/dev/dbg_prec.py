"""Debug (GPU box): per-point error pattern of the fused kernel vs the reference fixture's points."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_util as gu
from hip_util import setup
name = sys.argv[1] if len(sys.argv) > 1 else "full_dtu_ns3"
for prec in sys.argv[2:] or ["bf16", "fp16"]:
    fx, spec, net, rend = setup(name, precision=prec)
    xyz = torch.from_numpy(fx["pts_xyz_coarse"]).cuda(); vd = torch.from_numpy(fx["pts_dirs_coarse"]).cuda()
    for rep in range(2):
        out = net(xyz, coarse=True, viewdirs=vd).cpu().numpy()[0]
        ref = fx["pts_out_coarse"][0]
        err = np.abs(out[:, :3] - ref[:, :3]).max(-1)
        n = err.shape[0]
        print(f"{name} {prec} rep{rep}: max {err.max():.4f} mean {err.mean():.5f} frac>0.01 {np.mean(err > 0.01):.3f} nan {np.isnan(out).sum()}")
        e = np.pad(err, (0, (-n) % 128)).reshape(-1, 128)
        print("  by tile:", np.round(e.mean(1), 4))
        print("  by pos/8:", np.round(e.mean(0).reshape(16, 8).mean(1), 4))
