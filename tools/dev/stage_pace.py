"""Diagnostic: cycles per 16-MFMA stage of the fused kernel as a function of the stream size (n_blocks), to see whether
the stage pace follows the L2 residency of the weight stream."""
import sys, os, json
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import golden_util as gu
from hip_util import build_net, build_renderer
from pixel_nerf_multiscale_amd import util
for nb in (3, 4, 5, 6, 8):
    spec = dict(gu._BASE); spec.update(seed=100, d_hidden=512, lat=[(256, 8, 8)], NS=1, SB=1, image=(128, 128), focal=131.25, N=0,
                                       Kc=128, Kf=0, Kfd=0, n_blocks=nb, combine_layer=3, use_code_viewdirs=False,
                                       z_near=1.25, z_far=2.75, radius=2.0, white_bkgd=True, lindisp=False)
    poses = np.stack([gu.pose_spherical(0.0, -20.0, 2.0)])[None]
    net = build_net(spec, poses, "cuda", "bf16"); rend = build_renderer(spec)
    tgt = util.pose_spherical(75.0, -25.0, 2.0)[None].cuda()
    rays = util.gen_rays(tgt, 128, 128, torch.tensor(131.25), 1.25, 2.75).reshape(1, -1, 8).contiguous()
    for _ in range(3): rend(net, rays)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): rend(net, rays)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    stages = 3 + 3 * (4 + 1 + 65) + (nb - 3) * 65 + 2
    tiles_per_wg = 16384 / 256
    print(f"n_blocks={nb} stream={stages*16/1024:.2f} MB  {ms:.3f} ms/frame  us/stage={ms*1e3/(tiles_per_wg*stages):.4f}")
