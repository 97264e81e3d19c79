"""BASELINE.json configs 5 and 1 at their FULL sizes, through the C ABI on the GPU.

cfg 5 (`conf/exp/sn64_multiscale.conf:12-14,25,34,87-88` of the reference): 4-level latent [64@64x64, 64@64x64, 128@32x32,
256@16x16] of a 128x128 image with `use_first_pool = false`, 2 source views, `use_code_viewdirs`, `depth_std = 1.0`,
white background, the three points (Kc, Kf, Kfd) = (32,16,8) / (64,32,16) / (128,64,32) of the sample schedule — one
complete 16384-ray frame each.  The fused fp16 kernel (the dtype the config names, and bench.py's headline dtype) runs its
partially projected stream here: levels 0-2 gathered (one 256-channel group, cached between a view's blocks), the 16x16
level folded into lin_z over 256 texels.  The smaller fixtures exercise the same code with 16x16..4x4 / 32x32..8x8 maps;
this file is the shape the number in bench.py's `secondary` is quoted on.

cfg 1: 64x64 render of the SRN chairs 1-view shape, 32 coarse samples, no fine pass — the reference's own CPU-runnable
case — fp32 HIP path against the oracle restatement on the whole 4096-ray frame with explicit noise.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle_util import maxdiff
from test_gpu_parity import FINE_E2E_FLOOR_DB, FLOOR_DB, _psnr, _staged_render

pytestmark = pytest.mark.gpu

CFG5_LAT = [(64, 64, 64), (64, 64, 64), (128, 32, 32), (256, 16, 16)]


def _cfg5_spec(Kc, Kf, Kfd):
    spec = dict(gu.CASES["full_ns1"])
    spec.update(seed=105, lat=CFG5_LAT, NS=2, SB=1, image=(128, 128), focal=131.25, N=0, Kc=Kc, Kf=Kf, Kfd=Kfd,
                depth_std=1.0, lindisp=False, white_bkgd=True, use_code_viewdirs=True, z_near=0.8, z_far=1.8, radius=1.3)
    return spec


def _frame(spec, theta=75.0, phi=-25.0):
    from pixel_nerf_multiscale_amd import util
    W, H = spec["image"]
    tgt = util.pose_spherical(theta, phi, spec["radius"])[None].cuda()
    return util.gen_rays(tgt, W, H, torch.tensor(spec["focal"]), spec["z_near"], spec["z_far"]).reshape(1, -1, 8).contiguous()


@pytest.mark.parametrize("Kc,Kf,Kfd", [(32, 16, 8), (64, 32, 16), (128, 64, 32)])
def test_cfg5_multiscale_full_size(Kc, Kf, Kfd):
    from hip_util import build_net, build_renderer
    spec = _cfg5_spec(Kc, Kf, Kfd)
    poses = np.stack([gu.pose_spherical(30.0 * v, -20.0, spec["radius"]) for v in range(2)])[None]
    rays = _frame(spec)
    assert rays.shape[1] == 128 * 128
    K = Kc + Kf
    outs, nets = {}, {}
    for p in ("fp32", "fp16"):
        net = build_net(spec, poses, "cuda", p)
        assert net.resolved_precision(net.mlp_coarse, net.mlp_fine) == p
        if p == "fp16":      # the stream the bench measures: 256 texels of the last level projected, one gathered group
            v, _ = net.views_struct(p)
            m, _ = net.mlp_struct(net.mlp_coarse, p, v)
            assert m.packed_texels == 16 * 16 and net.d_latent == 512
        rend = build_renderer(spec)
        rend.forced_seed, rend.keep_samples = 2024, True
        outs[p] = rend(net, rays, want_weights=True)
        nets[p] = (net, rend)
    lo, hi = spec["z_near"], spec["z_far"]
    for p in ("fp32", "fp16"):
        o = outs[p]
        for lvl in ("coarse", "fine"):
            z, w = o[lvl].z[0], o[lvl].weights[0]
            assert not torch.isnan(o[lvl].rgb).any() and not torch.isnan(o[lvl].depth).any(), (p, lvl)
            assert bool((z[:, 1:] >= z[:, :-1]).all()), (p, lvl, "positions ascend")
            # importance samples have no upper clamp in the reference (nerf.py:138-143: bin index i can equal Kc, t = (i + r) / Kc
            # reaches 1 + 1 / Kc): a fine position may lie up to one coarse bin beyond `far`
            slack = (hi - lo) / Kc if lvl == "fine" else 0.0
            assert float(z.min()) >= lo - 1e-6 and float(z.max()) <= hi + slack + 1e-6, (p, lvl)
            assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-3, (p, lvl)
            # white background: rgb = sum w c + (1 - sum w) with c in [0, 1]
            assert float(o[lvl].rgb.min()) >= -1e-3 and float(o[lvl].rgb.max()) <= 1.0 + 1e-3, (p, lvl)
            assert float(o[lvl].depth.min()) >= 0.0 and float(o[lvl].depth.max()) <= hi + slack + 1e-3, (p, lvl)
        assert o.fine.z.shape[-1] == K and o.coarse.z.shape[-1] == Kc
    # the coarse positions come from the same counter-based draws in both precisions
    assert torch.equal(outs["fp16"].coarse.z, outs["fp32"].coarse.z)
    # precision: coarse pass, fine pass at the fp32 path's positions, fine pass end to end
    assert _psnr(outs["fp16"].coarse.rgb.cpu(), outs["fp32"].coarse.rgb.cpu()) >= FLOOR_DB["fp16"]
    z32 = outs["fp32"].fine.z
    xyz = (rays[:, :, None, :3] + z32[..., None] * rays[:, :, None, 3:6]).reshape(1, -1, 3).contiguous()
    vd = rays[:, :, None, 3:6].expand(-1, -1, K, -1).reshape(1, -1, 3).contiguous()
    inj = {}
    for p in ("fp32", "fp16"):
        net, rend = nets[p]
        pts = net(xyz, coarse=False, viewdirs=vd).reshape(-1, K, 4).contiguous()
        inj[p] = rend._composite_native(rays.reshape(-1, 8), z32.reshape(-1, K).contiguous(), pts)[1].cpu()
    assert _psnr(inj["fp32"], outs["fp32"].fine.rgb.cpu().reshape(-1, 3)) >= 90.0
    assert _psnr(inj["fp16"], inj["fp32"]) >= FLOOR_DB["fp16"], "fine pass at the fp32 sample positions"
    assert _psnr(outs["fp16"].fine.rgb.cpu(), outs["fp32"].fine.rgb.cpu()) >= FINE_E2E_FLOOR_DB["fp16"], "fine, end to end"
    # the fused launches against the staged launches, bit for bit, on the whole frame
    net, rend = nets["fp16"]
    staged = _staged_render(net, rend, rays)
    for lvl, (w, rgb, depth, z) in staged.items():
        assert torch.equal(outs["fp16"][lvl].z.reshape(-1, z.shape[1]), z), (lvl, "z")
        assert torch.equal(outs["fp16"][lvl].weights.reshape(-1, w.shape[1]), w), (lvl, "weights")
        assert torch.equal(outs["fp16"][lvl].rgb.reshape(-1, 3), rgb), (lvl, "rgb")
        assert torch.equal(outs["fp16"][lvl].depth.reshape(-1), depth), (lvl, "depth")
    # two half-frame calls keyed by the global ray index == the frame (what a 2-rank shard renders)
    half = rays.shape[1] // 2
    rend.ray_index_base = 0
    a = rend(net, rays[:, :half].contiguous())
    rend.ray_index_base = half
    b = rend(net, rays[:, half:].contiguous())
    rend.ray_index_base = 0
    assert torch.equal(torch.cat([a.fine.rgb, b.fine.rgb], 1), outs["fp16"].fine.rgb)


def test_cfg1_srn_64x64_k32_fp32_path_matches_the_oracle():
    """BASELINE cfg 1's exact shape: 4096 rays x 32 coarse samples, 1 view, 256 x 8 x 8 latent of a 128 x 128 source image,
    white background, z in [1.25, 2.75] — fp32 HIP path vs the oracle restatement (the fixtures pin both to the reference
    on 16-ray samples of wider shapes; this is the whole frame of the shape BASELINE.json names as CPU-runnable)."""
    from hip_util import build_net, build_renderer
    from oracle import pixelnerf_oracle as orc
    spec = dict(gu.CASES["full_ns1"])
    spec.update(seed=101, lat=[(256, 8, 8)], NS=1, image=(128, 128), focal=131.25, Kc=32, Kf=0, Kfd=0, N=0,
                z_near=1.25, z_far=2.75, radius=2.0, white_bkgd=True, lindisp=False)
    poses = gu.pose_spherical(0.0, -20.0, spec["radius"])[None, None]
    from pixel_nerf_multiscale_amd import util
    tgt = util.pose_spherical(75.0, -25.0, spec["radius"])[None]
    # a 64 x 64 render of the 128 x 128 camera: half the focal length and image size
    rays = util.gen_rays(tgt, 64, 64, torch.tensor(spec["focal"] * 0.5), spec["z_near"], spec["z_far"]).reshape(1, -1, 8).contiguous()
    assert rays.shape[1] == 4096
    g = torch.Generator().manual_seed(1)
    noise = dict(noise_c=torch.rand(4096, 32, generator=g))
    W, H = spec["image"]
    cam = orc.encode_cameras(torch.from_numpy(poses), spec["focal"], None, W, H)
    sd_c = {k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, "coarse").items()}
    with torch.no_grad():
        ref = orc.render(sd_c, None, cam, [torch.from_numpy(x) for x in gu.make_latents(spec)], rays, 1, 32, 0, 0,
                         spec["depth_std"], True, False, noise, use_code_viewdirs=False)
    net = build_net(spec, poses, "cuda", "fp32")
    rend = build_renderer(spec)
    rend.fixed_noise = {k: v.cuda() for k, v in noise.items()}
    out = rend(net, rays.cuda(), want_weights=True)
    assert maxdiff(out.coarse.rgb.cpu(), ref["coarse"]["rgb"]) <= 1e-4
    assert maxdiff(out.coarse.depth.cpu(), ref["coarse"]["depth"]) <= 1e-4 * (spec["z_far"] - spec["z_near"])
    assert maxdiff(out.coarse.weights.cpu(), ref["coarse"]["weights"]) <= 1e-4
    # and the fused 16-bit kernels on the same frame against it (headline dtype at SURVEY 8c's bound)
    for p in ("fp16", "bf16"):
        net16 = build_net(spec, poses, "cuda", p)
        o16 = rend(net16, rays.cuda())
        assert _psnr(o16.coarse.rgb.cpu(), ref["coarse"]["rgb"]) >= FLOOR_DB[p], p


@pytest.mark.parametrize("NS", [2, 3, 4])
@pytest.mark.parametrize("combine", ["average", "max"])
def test_multiview_reduction_against_the_fp32_path_with_the_views_permuted(NS, combine):
    """util.combine_interleaved (util.py:466-476) reduces fp32 activations, symmetrically in the view order.  The fused kernel
    parks the streams of views 0 .. NS-2 in its 16-bit format (the last view's stays in registers), so the view order is
    not symmetric in the last bits — pinned here: coarse-pass PSNR against the fp32 path for 2, 3 and 4 views, mean and max,
    in the given order AND with the views rotated; with net.park_precision = "fp32" (pnr_params.park_fp32) the streams wait as
    fp32 and the result is at least as close.  """
    from hip_util import build_net, build_renderer
    spec = dict(gu.CASES["full_ns3"])
    spec.update(NS=NS, combine_type=combine, seed=61 + NS, Kc=48, Kf=0, Kfd=0)
    poses = np.stack([gu.pose_spherical(25.0 * v, -20.0, spec["radius"]) for v in range(NS)])[None]
    W, H = spec["image"]
    g = torch.Generator().manual_seed(4)
    tgt = gu.pose_spherical(70.0, -25.0, spec["radius"])
    rays = torch.from_numpy(gu.pinhole_rays(tgt, W, H, spec["focal"], spec["z_near"], spec["z_far"],
                                            torch.randperm(W * H, generator=g)[:1500].numpy()))[None].cuda()
    lat = gu.make_latents(spec)

    def render(prec, perm, park="16bit"):
        net = build_net(spec, poses[:, perm], "cuda", prec)
        net.encoder.set_latents([torch.from_numpy(m[perm]).cuda() for m in lat])
        net.park_precision = park
        rend = build_renderer(spec)
        rend.forced_seed = 8
        return rend(net, rays).coarse.rgb.cpu()

    ident = list(range(NS))
    rot = ident[1:] + ident[:1]
    ref = render("fp32", ident)
    assert maxdiff(render("fp32", rot), ref) <= 2e-5                 # the fp32 path does not care about the order
    for p in ("fp16", "bf16"):
        a, b = render(p, ident), render(p, rot)
        assert _psnr(a, ref) >= FLOOR_DB[p] and _psnr(b, ref) >= FLOOR_DB[p], (p, _psnr(a, ref), _psnr(b, ref))
        a32, b32 = render(p, ident, "fp32"), render(p, rot, "fp32")
        assert _psnr(a32, ref) >= FLOOR_DB[p] and _psnr(b32, ref) >= FLOOR_DB[p]
        # fp32 parks: the reduction itself is exact fp32, so rotating the views moves the render by less than the 16-bit parks do
        assert _psnr(a32, b32) >= _psnr(a, b) - 1.0, (p, _psnr(a32, b32), _psnr(a, b))
