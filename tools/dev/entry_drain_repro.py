"""Diagnostic (GPU box): run-to-run determinism and parity of the fp16 / bf16 kernels vs the fp32 path in rays mode — the script
behind the statement-entry drain evidence of DESIGN.md §4.1 (tools/dev/build_variant.sh bad "noentrydrain" builds the failing library, ... bad_a "noentrydrain drainA" the
repaired one; run with PNR_LIB=tools/dev/libpnr_<NAME>.so)."""
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
