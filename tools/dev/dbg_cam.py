import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, numpy as np
from hip_util import setup
from pixel_nerf_multiscale_amd import util
name, prec = sys.argv[1], sys.argv[2]
fx, spec, net, rend = setup(name, precision=prec)
rend.fixed_noise = None
rend.forced_seed = 31
rend.keep_samples = True
pose = util.pose_spherical(40.0, -30.0, spec["radius"])
W, H, c = 24, 17, None
f = (spec["focal"], spec["focal"] * 1.1)
m = [float(x) for x in pose.flatten().tolist()]
cam = (m, W, H, float(f[0]), float(f[1]), W * 0.5, H * 0.5, float(spec["z_near"]), float(spec["z_far"]), 0, W * H)
a = rend._forward_fused(net, None, True, camera=cam)
rays = util.gen_rays_device(pose, W, H, f, spec["z_near"], spec["z_far"], c=c, device="cuda")
b = rend(net, rays[None], want_weights=True)
for k in ("z", "weights", "rgb", "depth"):
    x, y = a.coarse[k].reshape(W * H, -1), b.coarse[k].reshape(W * H, -1)
    d = (x - y).abs()
    bad = (d > 0).nonzero()
    print(k, "max", float(d.max()), "n_bad", len(bad), "first", bad[:5].tolist())
    if k == "z" and len(bad):
        i, j = bad[0].tolist()
        print("  z", x[i, j].item(), y[i, j].item(), "ray", rays[i].tolist())
