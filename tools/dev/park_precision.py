"""Diagnostic (GPU box): PSNR of the low-precision kernel vs the fp32 path on multi-view shapes — coarse pass, fine pass end
to end — for the library named by PNR_LIB (compare a 16-bit-park build with an fp32-park build)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_util as gu
from hip_util import build_net, build_renderer

def psnr(a, b):
    m = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if m == 0 else -10 * np.log10(m)

for NS, comb, lat, cv, lindisp in ((3, "max", [(256, 8, 8)], False, False), (3, "average", [(256, 8, 8)], False, False),
                                   (2, "average", [(256, 8, 8)], False, False), (4, "average", [(256, 8, 8)], True, False),
                                   (3, "average", [(256, 19, 25)], False, True),
                                   (2, "average", [(64, 32, 32), (64, 32, 32), (128, 16, 16), (256, 8, 8)], True, False)):
    for seed in (71, 72, 73):
        spec = dict(gu.CASES["full_ns1"]); spec.update(NS=NS, SB=1, N=4000, use_code_viewdirs=cv, seed=seed, combine_type=comb, lat=lat,
                                                       lindisp=lindisp, Kc=64, Kf=32, Kfd=16)
        if lindisp:
            spec.update(z_near=0.1, z_far=5.0, white_bkgd=False, image=(400, 300), focal=360.0)
        rays_np, poses = gu.make_inputs(spec)
        rays = torch.from_numpy(rays_np).cuda()
        outs = {}
        for p in ("fp32", "fp16", "bf16"):
            net = build_net(spec, poses, "cuda", p)
            rend = build_renderer(spec); rend.forced_seed = 5
            o = rend(net, rays)
            outs[p] = (o.coarse.rgb.cpu(), o.fine.rgb.cpu())
        print(f"NS {NS} {comb:7s} L {sum(c for c,_,_ in lat)} T {lat[-1][1]*lat[-1][2]:3d} seed {seed}: " +
              "  ".join(f"{p} coarse {psnr(outs[p][0], outs['fp32'][0]):5.1f} fine-e2e {psnr(outs[p][1], outs['fp32'][1]):5.1f}" for p in ("fp16", "bf16")), flush=True)
