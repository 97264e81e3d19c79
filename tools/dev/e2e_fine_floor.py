import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import golden_util as gu
from hip_util import build_net, build_renderer
def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean()); return 99.0 if mse == 0 else -10 * np.log10(mse)
for NS, SB in ((1, 3), (1, 1), (2, 2)):
    for seed in (70 + NS + SB, 5, 6):
        spec = dict(gu.CASES["full_ns1"]); spec.update(NS=NS, SB=SB, N=2000, seed=seed)
        rays_np, poses = gu.make_inputs(spec)
        rays = torch.from_numpy(rays_np).cuda()
        res = {}
        for tag, p, proj in (("fp32", "fp32", False), ("proj", "bf16", True), ("gen", "bf16", False), ("proj16", "fp16", True), ("gen16", "fp16", False)):
            net = build_net(spec, poses, "cuda", p); net.project_latent = proj
            rend = build_renderer(spec); rend.forced_seed = 5
            o = rend(net, rays)
            res[tag] = (o.coarse.rgb.cpu(), o.fine.rgb.cpu())
        line = f"NS {NS} SB {SB} seed {seed}:"
        for tag in ("proj", "gen", "proj16", "gen16"):
            line += f"  {tag} coarse {psnr(res[tag][0], res['fp32'][0]):.1f} fine-e2e {psnr(res[tag][1], res['fp32'][1]):.1f}"
        print(line, flush=True)
