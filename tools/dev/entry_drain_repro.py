"""Diagnostic (GPU box): run-to-run determinism and parity of the fp16 / bf16 kernels vs the fp32 path in rays mode — the script
behind round 2's statement-entry experiments.  The cause is known now (DESIGN.md 4.1 "Statement-entry rule",
profiles/r03_entry_hazard_isa_excerpt.txt: a pending compiler reload landing in a register the statement had already written); the
generator enforces the entry guard, so there is no failing variant to build any more.  Kept as a determinism check for PNR_LIB builds."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_util as gu
from hip_util import build_net, build_renderer
for name, n_rays in (("full_ns1", 24), ("full_ns1", 2000), ("full_ns3", 24), ("full_ns3", 1500), ("full_dtu_ns3", 16), ("full_dtu_ns3", 500)):
    spec = dict(gu.CASES[name]); spec.update(N=n_rays, Kf=0, Kfd=0)
    rays_np, poses = gu.make_inputs(spec)
    rays = torch.from_numpy(rays_np).cuda()
    outs = {}
    for p in ("fp32", "fp16", "bf16"):
        net = build_net(spec, poses, "cuda", p)
        rend = build_renderer(spec); rend.forced_seed = 5
        rs = [rend(net, rays).coarse.rgb.cpu().numpy() for _ in range(3)]
        outs[p] = rs
    for p in ("fp16", "bf16"):
        mse = float(((outs[p][0] - outs["fp32"][0]) ** 2).mean())
        det = max(np.abs(outs[p][0] - o).max() for o in outs[p][1:])
        print(f"{name} rays {n_rays:5d} (K={spec['Kc']}) {p}: psnr {-10*np.log10(mse+1e-30):.1f} dB rep-to-rep {det:.1e}", flush=True)
