"""debug: render_image(seed) vs chunked forward(ray_index_base, seed) on the same encode, one process"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from test_00_ranks_on_one_card import _eval_setup
from pixel_nerf_multiscale_amd import util
from pixel_nerf_multiscale_amd.parallel import frame_seed
net, rend, data = _eval_setup()
d = data[1]
W = H = 32
dev = "cuda"
poses = torch.as_tensor(d["poses"]).float()
focal = torch.tensor(d["focal"])
with torch.no_grad():
    net.encode(d["images"][:1].to(dev).unsqueeze(0), poses[:1].to(dev).unsqueeze(0), focal[None].to(dev))
    for vi in (1, 2):
        seed = frame_seed(99, vi)
        rend.forced_seed = seed
        rgb, depth = rend.render_image(net, poses[vi], W, H, focal, 1.25, 2.75)
        rays = util.gen_rays_device(poses[vi], W, H, focal, 1.25, 2.75, device=dev)
        full = rend(net, rays[None]).fine.rgb[0]
        print("view", vi, "camera vs ray tensor max diff", float((rgb.reshape(-1, 3) - full).abs().max()))
        parts, at = [], 0
        for r in torch.split(rays, 300, dim=0):
            for lo, hi in ((0, (r.shape[0] + 1) // 2), ((r.shape[0] + 1) // 2, r.shape[0])):
                rend.ray_index_base = at + lo
                parts.append(rend(net, r[lo:hi][None].contiguous()).fine.rgb[0])
            at += r.shape[0]
        rend.ray_index_base = 0
        ch = torch.cat(parts, 0)
        print("view", vi, "chunked+sharded vs full max diff", float((ch - full).abs().max()), "n", ch.shape[0])
        # and through the wrapper the evaluate() loop uses (world 1: no collective)
        from pixel_nerf_multiscale_amd.render.nerf import _ShardedRenderWrapper
        rp = _ShardedRenderWrapper(net, rend, simple_output=True).eval()
        parts, at = [], 0
        for r in torch.split(rays, 300, dim=0):
            parts.append(rp(r[None], ray_index_base=at, seed=seed)[0][0])
            at += r.shape[0]
        print("view", vi, "wrapper chunks vs full max diff", float((torch.cat(parts, 0) - full).abs().max()))
