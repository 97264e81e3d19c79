"""Debug (GPU box): fp16/bf16 fused kernel vs fixtures, several cases, projected on/off, 3 reps (determinism)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_util as gu
from hip_util import setup
for name in (sys.argv[1:] or ["full_ns1", "full_ns3", "full_multiscale_ns2", "full_dtu_ns3"]):
    for prec in ["fp16", "bf16"]:
        for proj in (True, False):
            fx, spec, net, rend = setup(name, precision=prec)
            net.project_latent = proj
            xyz = torch.from_numpy(fx["pts_xyz_coarse"]).cuda(); vd = torch.from_numpy(fx["pts_dirs_coarse"]).cuda()
            outs = [net(xyz, coarse=True, viewdirs=vd).cpu().numpy()[0] for _ in range(3)]
            ref = fx["pts_out_coarse"][0]
            err = np.abs(outs[0][:, :3] - ref[:, :3]).max(-1)
            det = max(np.abs(outs[0] - o).max() for o in outs[1:])
            e = np.pad(err, (0, (-len(err)) % 128)).reshape(-1, 128).mean(0).reshape(16, 8).mean(1)
            print(f"{name:20s} {prec} proj={int(proj)}: max {err.max():.4f} mean {err.mean():.5f} rep-to-rep {det:.2e}  by pos/8 {np.round(e, 3)}", flush=True)
