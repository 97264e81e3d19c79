"""Parity tests proper: the HIP path (through the C ABI) against the golden fixtures produced by the
reference and against the oracle restatement, on the same inputs and the same recorded noise.
Tolerances: fp32 path |d rgb|, |d w| <= 1e-4, |d depth| <= 1e-4*(far-near)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import pixelnerf_oracle as orc
from oracle_util import maxdiff, noise_from_fixture, oracle_points_f64, oracle_render

pytestmark = pytest.mark.gpu
ALL = sorted(gu.CASES)
TINY = [n for n in ALL if n.startswith("tiny")]


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("name", ALL)
def test_render_fp32_matches_reference(name):
    from hip_util import setup
    fx, spec, net, rend = setup(name, precision="fp32")
    out = rend(net, _dev(fx["rays"]), want_weights=True)
    span = max(spec["z_far"] - spec["z_near"], 1.0)
    for lvl in ("coarse", "fine"):
        if lvl == "fine" and spec["Kf"] == 0:
            assert "fine" not in out
            continue
        assert maxdiff(out[lvl].rgb.cpu(), fx[f"{lvl}_rgb"]) <= 1e-4, lvl
        assert maxdiff(out[lvl].weights.cpu(), fx[f"{lvl}_weights"]) <= 1e-4, lvl
        assert maxdiff(out[lvl].depth.cpu(), fx[f"{lvl}_depth"]) <= 1e-4 * span * 4, lvl


@pytest.mark.parametrize("name", ALL)
def test_point_mlp_fp32_matches_reference(name):
    """PixelNeRFNet.forward (explicit points) == reference model output on the reference's own points."""
    from hip_util import setup
    fx, spec, net, rend = setup(name, precision="fp32")
    for tag in ("coarse", "fine"):
        if f"pts_xyz_{tag}" not in fx:
            continue
        out = net(_dev(fx[f"pts_xyz_{tag}"]), coarse=(tag == "coarse"), viewdirs=_dev(fx[f"pts_dirs_{tag}"]))
        o, ref = out.cpu().numpy(), fx[f"pts_out_{tag}"]
        assert np.abs(o[..., :3] - ref[..., :3]).max() <= 1e-4, tag                       # rgb
        # sigma = a 512-term sum with O(50) partial sums scaled x20 by the synthetic lin_out: fp32 summation-order
        # noise alone is ~5e-4 absolute; it enters the render as delta*sigma with delta ~ 0.02
        assert (np.abs(o[..., 3] - ref[..., 3]) <= 1e-3 + 1e-4 * np.abs(ref[..., 3])).all(), tag
        # ... and that is what it is: against the float64 recomputation of the same points the HIP result sits in the
        # same band as the reference's own fp32 result (measured |ref - f64| on sigma: 5e-5 … 4e-4, 1.2e-3 on the DTU
        # case whose coordinates reach 40 units) — both fp32 sums bracket the float64 value
        t64 = oracle_points_f64(fx, tag)
        e_ref, e_hip = np.abs(ref - t64), np.abs(o - t64)
        assert e_hip[..., 3].max() <= 4.0 * max(e_ref[..., 3].max(), 1e-4), (tag, e_hip[..., 3].max(), e_ref[..., 3].max())
        assert e_hip[..., :3].max() <= 4.0 * max(e_ref[..., :3].max(), 1e-5), (tag, e_hip[..., :3].max(), e_ref[..., :3].max())


@pytest.mark.parametrize("name", TINY)
def test_stage_kernels_match_oracle(name):
    from hip_util import setup
    fx, spec, net, rend = setup(name)
    res = oracle_render(fx)
    rays = _dev(fx["rays"]).reshape(-1, 8)
    # a2 sample_coarse
    z = rend.sample_coarse(rays)
    assert maxdiff(z.cpu(), res["coarse"]["z"]) <= 2e-6 * max(1.0, spec["z_far"])
    # a4 composite on the oracle's model outputs
    zc = res["coarse"]["z"].cuda()
    w, rgb, depth = rend._composite_native(rays, zc.contiguous(), res["coarse"]["pts_out"].cuda().contiguous())
    assert maxdiff(w.cpu(), res["coarse"]["weights"].reshape(w.shape)) <= 2e-6
    assert maxdiff(rgb.cpu(), res["coarse"]["rgb"].reshape(rgb.shape)) <= 5e-6
    assert maxdiff(depth.cpu(), res["coarse"]["depth"].reshape(-1)) <= 2e-5
    # a5-a7 fine sampling + sort from the oracle's coarse weights/depth
    if spec["Kf"] > 0:
        rend.last_seed = 0
        zf = rend.sample_fine_sorted(rays, zc, res["coarse"]["weights"].reshape(w.shape).cuda(),
                                     res["coarse"]["depth"].reshape(-1).cuda())
        assert maxdiff(zf.cpu(), res["fine"]["z"]) <= 5e-6 * max(1.0, spec["z_far"])
        assert bool((zf[:, 1:] >= zf[:, :-1]).all())


def test_generic_model_protocol():
    """The renderer still drives an arbitrary Python model (reference protocol nerf.py:188,212-216)."""
    from hip_util import setup
    fx, spec, net, rend = setup("tiny_ns1")

    class Toy(torch.nn.Module):
        use_viewdirs = True

        def forward(self, xyz, coarse=True, viewdirs=None):
            s = torch.sin(xyz * 3.0).abs()
            sig = (xyz.norm(dim=-1, keepdim=True) < 1.0).float() * (8.0 if coarse else 11.0) + viewdirs[..., :1].abs()
            return torch.cat([s, sig], dim=-1)

    toy = Toy()
    rays = _dev(fx["rays"])
    out = rend(toy, rays, want_weights=True)
    # oracle: same stages with the toy model on CPU
    n = noise_from_fixture(fx)
    r = torch.from_numpy(fx["rays"]).reshape(-1, 8)
    zc = orc.sample_coarse(r, spec["Kc"], spec["lindisp"], n["noise_c"])

    def run(z, coarse):
        K = z.shape[1]
        pts = (r[:, None, :3] + z.unsqueeze(2) * r[:, None, 3:6]).reshape(1, -1, 3)
        dirs = r[:, None, 3:6].expand(-1, K, -1).reshape(1, -1, 3)
        return orc.composite(r, z, toy(pts, coarse=coarse, viewdirs=dirs).reshape(-1, K, 4), spec["white_bkgd"])

    w, rgb, depth = run(zc, True)
    assert maxdiff(out.coarse.rgb.cpu().reshape(-1, 3), rgb) <= 1e-5
    zf = torch.sort(torch.cat([zc, orc.sample_fine(r, w, spec["Kc"], spec["lindisp"], n["u"], n["r"]),
                               orc.sample_fine_depth(r, depth, spec["depth_std"], n["g"])], -1), -1)[0]
    w2, rgb2, depth2 = run(zf, False)
    assert maxdiff(out.fine.rgb.cpu().reshape(-1, 3), rgb2) <= 1e-5
    assert maxdiff(out.fine.weights.cpu().reshape(w2.shape), w2) <= 1e-5


def test_kernel_rng_sharding_is_bit_identical():
    """Counter-based noise keyed by the global ray index: rendering a frame in two shards (ray_index_base)
    gives exactly the bytes of the unsharded frame."""
    from hip_util import setup
    fx, spec, net, rend = setup("tiny_ns2_codeview")
    rend.fixed_noise = None
    rays = _dev(fx["rays"])
    torch.manual_seed(5)
    full = rend(net, rays)
    seed = rend.last_seed
    h = rays.shape[1] // 2
    parts = []
    for lo, hi in ((0, h), (h, rays.shape[1])):
        torch.manual_seed(5)
        rend.ray_index_base = lo
        parts.append(rend(net, rays[:, lo:hi].contiguous()))
        assert rend.last_seed == seed
    rend.ray_index_base = 0
    assert torch.equal(torch.cat([p.fine.rgb for p in parts], 1), full.fine.rgb)
    assert torch.equal(torch.cat([p.fine.depth for p in parts], 1), full.fine.depth)


def test_kernel_rng_statistics():
    from pixel_nerf_multiscale_amd import NeRFRenderer
    rend = NeRFRenderer(n_coarse=64).cuda()
    rays = torch.zeros(4096, 8, device="cuda")
    rays[:, 6], rays[:, 7] = 0.0, 1.0
    z = rend.sample_coarse(rays, seed=1234)
    t = (z * 64 - torch.arange(64, device="cuda")[None]).flatten()      # the U[0,1) jitter
    assert 0.0 <= float(t.min()) and float(t.max()) < 1.0 + 1e-4
    assert abs(float(t.mean()) - 0.5) < 5e-3 and abs(float(t.var()) - 1 / 12) < 5e-3
    z2 = rend.sample_coarse(rays, seed=1234)
    assert torch.equal(z, z2)
    assert not torch.equal(z, rend.sample_coarse(rays, seed=1235))


def test_full_size_properties():
    """BASELINE-size frame (128x128 rays, 64+32 samples): size-independent properties of the stages."""
    from pixel_nerf_multiscale_amd import NeRFRenderer
    N, Kc, Kf, Kfd = 16384, 64, 32, 16
    g = torch.Generator(device="cuda").manual_seed(3)
    rays = torch.zeros(N, 8, device="cuda")
    rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(N, 3, device="cuda", generator=g), dim=-1)
    rays[:, 6], rays[:, 7] = 1.25, 2.75
    rend = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, white_bkgd=True).cuda()
    z = rend.sample_coarse(rays, seed=7)
    assert bool((z[:, 1:] > z[:, :-1]).all()) and float(z.min()) >= 1.25 and float(z.max()) <= 2.75
    out = torch.rand(N, Kc, 4, device="cuda", generator=g)
    out[..., 3] *= 30.0
    out[..., :3] = 1.0                                    # constant white colour
    w, rgb, depth = rend._composite_native(rays, z, out)
    assert float((rgb - 1.0).abs().max()) <= 1e-5         # sum w*1 + (1 - sum w) == 1
    assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
    assert bool(((depth >= 0) & (depth <= 2.75 + 1e-4)).all())
    zf = rend.sample_fine_sorted(rays, z, w, depth, seed=7)
    assert zf.shape == (N, Kc + Kf) and bool((zf[:, 1:] >= zf[:, :-1]).all())
    assert float(zf.min()) >= 1.25 - 1e-5 and float(zf.max()) <= 2.75 + 1.5 / Kc + 1e-4   # i may equal Kc (a5)
    # the merged set contains every coarse sample
    merged = torch.sort(torch.cat([z, zf], -1), -1)[0]
    assert merged.shape[1] == 2 * Kc + Kf


def test_empty_and_error_paths():
    from hip_util import setup
    fx, spec, net, rend = setup("tiny_ns1")
    wrapped = rend.bind_parallel(net, simple_output=True)
    rgb, depth = wrapped(torch.zeros(0, 8, device="cuda"))
    assert rgb.shape == (0, 3) and depth.shape == (0,)
    with pytest.raises(RuntimeError):
        rend(net, torch.from_numpy(fx["rays"]))           # CPU rays: no CPU path
    with pytest.raises(ValueError):
        rend(net, torch.zeros(3, 4, 8, device="cuda"))    # 3 objects vs 1 encoded


# ----------------------------------------------------------------------------- fused MFMA path (bf16 / fp16)
FULL = [n for n in ALL if n.startswith("full")]


def _psnr(a, b):
    mse = float(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).mean())
    return 99.0 if mse == 0 else -10.0 * np.log10(mse)


# Low-precision bound — ONE statement, enforced by every test below: the reference has no bf16/fp16
# numerics, so the bound is PSNR(low-precision render, fp32 render of the same rays with the same noise) on the rendered
# pixels of EITHER pass:   bf16 >= 42.4 dB   (the north-star's 0.05 dB budget for an uncorrelated error),
#                          fp16 >= 58 dB     (what precision="auto" selects).
# Point level (the network's rgb before compositing) both kernels are shape-independent: bf16 >= 52 dB, fp16 >= 68 dB
# (measured 60-62 / 78-79 dB on SRN, NMR and DTU shapes alike, tools/dev/bf16_gap.py); what differs between shapes is the
# compositing stage's gain on the sigma error (DTU: disparity steps of up to 2 units -> 44-45 dB in bf16).
# The end-to-end FINE pass draws its sample positions from the low-precision coarse weights: a cdf entry moving across
# a draw u makes that importance sample jump a bin (SURVEY §8c caveat; the reference's own CPU and GPU cumsum differ the
# same way).  That makes the end-to-end fine PSNR a heavy-tailed quantity, not a precision measure: over shapes x seeds x
# weight streams it spreads 41.9 .. 72 dB in bf16 and 54 .. 85 dB in fp16 (tools/dev/e2e_fine_floor.py), the projected
# and the general stream trading places by +-3 dB from seed to seed.  The precision statement for the fine pass is the
# one at INJECTED fp32 sample positions (the bound above, also checked against the reference's own fine pass in
# test_mfma_matches_reference).  End to end (a statistic of a discontinuous map: one flipped bin in one ray moves a
# 150-ray PSNR by several dB, so the floors are meant for samples of >= ~1000 rays; the 16-ray fixtures clear them too):
#   fp16: >= 50 dB — SURVEY §8(c)'s adopted bound holds end to end as well (measured minimum 54 dB);
#   bf16: >= 40 dB — BELOW the §8(c) bound, bf16 only: a reported sanity floor under the measured spread (41.9 dB minimum),
#         not a precision claim; bf16 is the dtype BASELINE.json names for cfg 2, fp16 is what precision="auto" selects.
# Round 4: ONE 16-bit dtype carries both the headline rate and the parity claim — fp16 (bench.HEADLINE_DTYPE, what
# precision="auto" selects).  Its floors are SURVEY §8(c)'s bound or tighter on every test of this suite
# (tests/test_host_cpu.py::test_headline_dtype_is_held_to_survey_8c checks exactly that on the CPU).  bf16 stays a
# supported, benchmarked secondary format with the floors below, labelled NON-CONFORMING: it does not meet §8(c) on
# every shape and nothing in the bench's headline is quoted on it.
SURVEY_8C_DB = 50.0
BF16_FLOOR_DB, FP16_FLOOR_DB = 42.4, 58.0
FLOOR_DB = {"bf16": BF16_FLOOR_DB, "fp16": FP16_FLOOR_DB}
FINE_E2E_FLOOR_DB = {"bf16": 40.0, "fp16": 50.0}
NON_CONFORMING_DTYPES = ("bf16",)


@pytest.mark.parametrize("prec,floor_pts,floor_px", [("bf16", 52.0, BF16_FLOOR_DB), ("fp16", 68.0, FP16_FLOOR_DB)])
@pytest.mark.parametrize("name", FULL)
def test_mfma_matches_reference(name, prec, floor_pts, floor_px):
    from hip_util import setup
    fx, spec, net, rend = setup(name, precision=prec)
    assert net.resolved_precision() == prec
    levels = ("coarse", "fine") if spec["Kf"] > 0 else ("coarse",)
    for tag in levels:
        out = net(_dev(fx[f"pts_xyz_{tag}"]), coarse=(tag == "coarse"), viewdirs=_dev(fx[f"pts_dirs_{tag}"])).cpu().numpy()
        ref = fx[f"pts_out_{tag}"]
        assert not np.isnan(out).any()
        assert _psnr(out[..., :3], ref[..., :3]) >= floor_pts, tag
        rel = np.abs(out[..., 3] - ref[..., 3]) / (1.0 + np.abs(ref[..., 3]))
        assert rel.max() <= (0.25 if prec == "bf16" else 0.05), tag          # sigma logits are x20 in the fixtures
    out = rend(net, _dev(fx["rays"]), want_weights=True)
    for lvl in levels:
        assert _psnr(out[lvl].rgb.cpu(), fx[f"{lvl}_rgb"]) >= (floor_px if lvl == "coarse" else FINE_E2E_FLOOR_DB[prec]), lvl
        # fine pass: its sample positions are drawn from the LOW-PRECISION coarse weights/depth, so a sample can land in
        # a neighbouring bin and move one ray's weights discontinuously (SURVEY §8c caveat) — looser bound there
        tol_w = {("bf16", "coarse"): 0.05, ("bf16", "fine"): 0.15, ("fp16", "coarse"): 0.01, ("fp16", "fine"): 0.03}[(prec, lvl)]
        assert maxdiff(out[lvl].weights.cpu(), fx[f"{lvl}_weights"]) <= tol_w, lvl
    if spec["Kf"] > 0:
        # the fine pass at the REFERENCE's own sample positions ("compare fine pass with z_samp injected"): the
        # reference's fine points through this precision's fine MLP, composited by the stage kernel, against the
        # reference's fine pixels — the stated bound, no resampling in between.  z = (p - o) . d for unit d.
        rays = _dev(fx["rays"]).reshape(-1, 8)
        K = spec["Kc"] + spec["Kf"]
        xyz = _dev(fx["pts_xyz_fine"]).reshape(rays.shape[0], K, 3)
        z = ((xyz - rays[:, None, :3]) * rays[:, None, 3:6]).sum(-1).contiguous()
        _, rgb_ref, _ = rend._composite_native(rays, z, _dev(fx["pts_out_fine"]).reshape(-1, K, 4).contiguous())
        assert maxdiff(rgb_ref.cpu(), fx["fine_rgb"].reshape(-1, 3)) <= 1e-4        # the injected route reproduces the reference
        pts = net(_dev(fx["pts_xyz_fine"]), coarse=False, viewdirs=_dev(fx["pts_dirs_fine"])).reshape(-1, K, 4).contiguous()
        _, rgb_inj, _ = rend._composite_native(rays, z, pts)
        assert _psnr(rgb_inj.cpu(), fx["fine_rgb"].reshape(-1, 3)) >= floor_px, "fine pass at the reference's sample positions"


@pytest.mark.parametrize("prec,floor", [("bf16", 50.0), ("fp16", 65.0)])
def test_mfma_frame_psnr(prec, floor):
    """4096 rays x 64 coarse samples, NS=1: MFMA path vs the fp32 HIP path (itself pinned to the reference at
    1e-4) with identical in-kernel noise.  Also exercises the tail tile and a ray count that is not a tile multiple."""
    from hip_util import build_net, build_renderer
    import golden_util as gu
    spec = dict(gu.CASES["full_ns1"]); spec.update(Kc=64, Kf=0, Kfd=0)
    poses = np.stack([gu.pose_spherical(0.0, -20.0, spec["radius"])])[None]
    g = torch.Generator().manual_seed(1)
    W, H = spec["image"]
    tgt = gu.pose_spherical(75.0, -25.0, spec["radius"])
    rays = torch.from_numpy(gu.pinhole_rays(tgt, W, H, spec["focal"], spec["z_near"], spec["z_far"],
                                            torch.randperm(W * H, generator=g)[:4093].numpy()))[None].cuda()
    outs = {}
    for p in ("fp32", prec):
        net = build_net(spec, poses, "cuda", p)
        rend = build_renderer(spec)
        rend.forced_seed = 99
        outs[p] = rend(net, rays).coarse.rgb.cpu()
    assert _psnr(outs[prec], outs["fp32"]) >= floor


@pytest.mark.parametrize("prec,floor,floor_self", [("bf16", 50.0, 50.0), ("fp16", 65.0, 65.0)])
@pytest.mark.parametrize("lat,image,NS", [((256, 8, 8), (128, 128), 1), ((256, 5, 7), (80, 56), 1), ((256, 8, 16), (128, 64), 1),
                                          ((256, 8, 8), (64, 64), 3), ((256, 6, 6), (48, 48), 2), ((256, 16, 16), (128, 128), 1),
                                          ("multiscale", (64, 64), 2)])
def test_projected_stream_matches_general_path(prec, floor, floor_self, lat, image, NS):
    """One view + one small latent map: the stream carries W_z . Lat (pnr_pack_mlp_projected) and the kernel skips the
    gather.  Same inputs through the projected stream, the general (gather + lin_z) stream and the fp32 path; also a
    texel count that is not a multiple of 16 (5x7), a non-square map, and 2 / 3 source views (one stream copy per view)."""
    import ctypes as C
    from hip_util import build_net, build_renderer
    import golden_util as gu
    from pixel_nerf_multiscale_amd import _native as N
    cv = False
    if lat == "multiscale":        # 4-level encoder output: levels 0-2 (256 channels) gathered, the 16x16 level projected
        lats, cv = [(64, 32, 32), (64, 32, 32), (128, 16, 16), (256, 8, 8)], True
        lat = lats[-1]
    else:
        lats = [lat]
    spec = dict(gu.CASES["full_ns1"]); spec.update(Kc=48, Kf=16, Kfd=8, lat=lats, image=image, seed=41, NS=NS, use_code_viewdirs=cv)
    poses = np.stack([gu.pose_spherical(10.0 + 35.0 * v, -20.0, spec["radius"]) for v in range(NS)])[None]
    W, H = image
    g = torch.Generator().manual_seed(3)
    tgt = gu.pose_spherical(60.0, -25.0, spec["radius"])
    rays = torch.from_numpy(gu.pinhole_rays(tgt, W, H, spec["focal"], spec["z_near"], spec["z_far"],
                                            torch.randperm(W * H, generator=g)[:1500].numpy()))[None].cuda()
    outs = {}
    for tag, p, proj in (("fp32", "fp32", False), ("proj", prec, True), ("gen", prec, False)):
        net = build_net(spec, poses, "cuda", p)
        net.project_latent = proj
        if p != "fp32":
            v, _ = net.views_struct(p)
            m, _ = net.mlp_struct(net.mlp_coarse, p, v)
            assert m.packed_texels == (lat[1] * lat[2] if proj else 0)
        rend = build_renderer(spec)
        rend.forced_seed = 7
        rend.keep_samples = True
        o = rend(net, rays, want_weights=True)
        outs[tag] = o
    for lvl in ("coarse",):        # the fine pass resamples from low-precision weights: compare the coarse pass pixel-wise
        ref = outs["fp32"][lvl].rgb.cpu()
        assert _psnr(outs["proj"][lvl].rgb.cpu(), ref) >= floor
        assert _psnr(outs["gen"][lvl].rgb.cpu(), ref) >= floor
        assert _psnr(outs["proj"][lvl].rgb.cpu(), outs["gen"][lvl].rgb.cpu()) >= floor_self
    assert _psnr(outs["proj"].fine.rgb.cpu(), outs["fp32"].fine.rgb.cpu()) >= FINE_E2E_FLOOR_DB[prec]
    # re-encoding (new latent) must re-pack: same weights, different map -> different output, still right
    net = build_net(spec, poses, "cuda", prec)
    rend = build_renderer(spec); rend.forced_seed = 7
    a = rend(net, rays).coarse.rgb.clone()
    net.encoder.set_latents([m * 0.5 for m in net.encoder.level_maps()])
    b = rend(net, rays).coarse.rgb
    net32 = build_net(spec, poses, "cuda", "fp32")
    net32.encoder.set_latents([m * 0.5 for m in net32.encoder.level_maps()])
    rend32 = build_renderer(spec); rend32.forced_seed = 7
    assert _psnr(b.cpu(), rend32(net32, rays).coarse.rgb.cpu()) >= floor
    assert float((a - b).abs().max()) > 1e-3
    # a projected stream is tied to its map: the C ABI refuses it for other view sets
    v, _ = net.views_struct(prec)
    m, _ = net.mlp_struct(net.mlp_coarse, prec, v)
    assert m.packed_texels > 0
    prm = net.params_struct(None, prec)
    xyz = torch.zeros(2, 8, 3, device="cuda"); out = torch.empty(2, 8, 4, device="cuda")
    v2 = N.pnr_views.from_buffer_copy(v); v2.n_objs = 2
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    rc = N.lib.pnr_point_mlp(C.byref(prm), C.byref(m), C.byref(v2), None, None, 0, N.ptr(xyz), N.ptr(xyz), 16, 8, N.ptr(out),
                             ws.data_ptr(), ws.numel(), N.current_stream(xyz.device))
    assert rc in (-6, -2, -1)


def test_mfma_multiview_large():
    """NS=3 (view spill/reduce path) on enough points that every workgroup loops over several tiles."""
    from hip_util import build_net
    import golden_util as gu
    spec = dict(gu.CASES["full_ns3"])
    fx = gu.load_fixture("full_ns3")
    g = torch.Generator().manual_seed(2)
    xyz = ((torch.rand(1, 40000, 3, generator=g) - 0.5) * 1.6).cuda()
    vd = torch.nn.functional.normalize(torch.randn(1, 40000, 3, generator=g), dim=-1).cuda()
    ref = build_net(spec, fx["poses"], "cuda", "fp32")(xyz, viewdirs=vd).cpu().numpy()
    out = build_net(spec, fx["poses"], "cuda", "fp16")(xyz, viewdirs=vd).cpu().numpy()
    assert _psnr(out[..., :3], ref[..., :3]) >= 65.0
    out = build_net(spec, fx["poses"], "cuda", "bf16")(xyz, viewdirs=vd).cpu().numpy()
    assert _psnr(out[..., :3], ref[..., :3]) >= 50.0


def test_mfma_requires_packed_and_supported_shape():
    """The low-precision path fails loudly (no fallback) on shapes the fused kernel is not built for."""
    from hip_util import setup
    fx, spec, net, rend = setup("tiny_ns1", precision="bf16")
    with pytest.raises(ValueError):
        rend(net, _dev(fx["rays"]))
    net.precision = "auto"                      # auto -> fp32 HIP path for d_hidden=32
    assert net.resolved_precision() == "fp32"
    rend(net, _dev(fx["rays"]))


def test_gen_rays_kernel_matches_host():
    """N1: on-device ray generation == the host gen_rays mirror == the fixtures' pinhole model."""
    from pixel_nerf_multiscale_amd import util
    import golden_util as gu
    pose = util.pose_spherical(75.0, -25.0, 2.0)
    W, H, f = 40, 30, 45.0
    dev = util.gen_rays_device(pose, W, H, f, 1.25, 2.75).cpu()
    host = util.gen_rays(pose[None], W, H, torch.tensor(f), 1.25, 2.75).reshape(-1, 8)
    assert maxdiff(dev, host) <= 2e-6
    ref = gu.pinhole_rays(gu.pose_spherical(75.0, -25.0, 2.0), W, H, f, 1.25, 2.75, np.arange(W * H))
    assert maxdiff(dev, ref) <= 2e-6


def test_gen_rays_kernel_matches_reference_fixture():
    """N1 pinned: pnr_gen_rays == the reference's own util.gen_rays outputs (tests/golden/gen_rays.npz, written by
    tools/gen_golden_rays.py from reference util.py:118-148,243-281)."""
    from pixel_nerf_multiscale_amd import util
    for cs in gu.load_rays_fixture():
        c = None if cs["c"] is None else torch.from_numpy(cs["c"])
        for i in range(cs["poses"].shape[0]):
            dev = util.gen_rays_device(torch.from_numpy(cs["poses"][i]), cs["W"], cs["H"], torch.from_numpy(cs["focal"]),
                                       cs["z_near"], cs["z_far"], c=c).cpu().numpy()
            assert maxdiff(dev, cs["rays"][i].reshape(-1, 8)) <= 2e-6, (cs["name"], i)


@pytest.mark.parametrize("NS,SB,cv,comb", [(2, 2, False, "average"), (4, 1, True, "average"), (1, 3, False, "average"),
                                           (3, 1, False, "max")])
def test_mfma_vs_fp32_path_superbatch_and_views(NS, SB, cv, comb):
    """Full-width MFMA kernel vs the fp32 HIP path (pinned to the reference by the fixtures, incl. SB=2) on shapes
    the fixtures do not hold at d_hidden=512: several objects per call, 2/4 source views, coded viewdirs."""
    from hip_util import build_net, build_renderer
    import golden_util as gu
    # the reference's sample schedule (conf/default.conf:50-53: 64 coarse + 32 fine, 16 of them depth-guided): the end-to-end
    # fine statistic is about bins flipping, and with a toy 16-bin schedule one flip moves a sample by 1/16 of the ray
    spec = dict(gu.CASES["full_ns1"]); spec.update(NS=NS, SB=SB, N=2000, use_code_viewdirs=cv, seed=70 + NS + SB, combine_type=comb,
                                                   Kc=64, Kf=32, Kfd=16)
    rays_np, poses = gu.make_inputs(spec)
    rays = torch.from_numpy(rays_np).cuda()
    outs, fine_fixed = {}, {}
    z32 = None
    for p in ("fp32", "fp16", "bf16"):
        net = build_net(spec, poses, "cuda", p)
        rend = build_renderer(spec)
        rend.forced_seed = 5
        rend.keep_samples = True
        o = rend(net, rays, want_weights=True)
        outs[p] = (o.fine.rgb.cpu(), o.fine.weights.cpu(), o.coarse.rgb.cpu())
        if p == "fp32":
            z32 = o.fine.z                                               # (SB, N, Kc + Kf)
        # the fine pass at the fp32 path's sample positions (compare the fine pass with z_samp injected as well as
        # end-to-end): same points through this precision's fine MLP, composited by the stage kernel
        K = z32.shape[-1]
        xyz = (rays[:, :, None, :3] + z32[..., None] * rays[:, :, None, 3:6]).reshape(SB, -1, 3).contiguous()
        vd = rays[:, :, None, 3:6].expand(-1, -1, K, -1).reshape(SB, -1, 3).contiguous()
        pts = net(xyz, coarse=False, viewdirs=vd).reshape(-1, K, 4).contiguous()
        _, rgb_f, _ = rend._composite_native(rays.reshape(-1, 8), z32.reshape(-1, K).contiguous(), pts)
        fine_fixed[p] = rgb_f.cpu().reshape(SB, -1, 3)
    assert outs["fp32"][0].shape == (SB, 2000, 3)
    assert _psnr(fine_fixed["fp32"], outs["fp32"][0]) >= 90.0                  # the injected route reproduces the fp32 render
    for p in ("fp16", "bf16"):
        assert _psnr(outs[p][2], outs["fp32"][2]) >= FLOOR_DB[p], (p, "coarse")
        assert _psnr(fine_fixed[p], fine_fixed["fp32"]) >= FLOOR_DB[p], (p, "fine pass at the fp32 sample positions")
        # end to end the fine pass resamples from the low-precision coarse weights (bins can flip): sanity floor only
        assert _psnr(outs[p][0], outs["fp32"][0]) >= FINE_E2E_FLOOR_DB[p], (p, "fine, end to end")


def test_render_image_from_camera_equals_forward_on_host_rays():
    """N1: frame rendered from (pose, intrinsics) with on-device rays == forward() on host-generated rays, bit for bit
    with the same kernel seed; async pinned D2H returns the same bytes."""
    from hip_util import setup
    from pixel_nerf_multiscale_amd import util
    fx, spec, net, rend = setup("tiny_ns2_codeview")
    rend.fixed_noise = None
    W, H = spec["image"]
    pose = util.pose_spherical(75.0, -25.0, spec["radius"])
    rend.forced_seed = 11
    rgb, depth = rend.render_image(net, pose, W, H, spec["focal"], spec["z_near"], spec["z_far"])
    rays = util.gen_rays(pose[None], W, H, torch.tensor(spec["focal"]), spec["z_near"], spec["z_far"]).reshape(1, -1, 8).cuda()
    ref = rend(net, rays)
    assert rgb.shape == (H, W, 3) and depth.shape == (H, W)
    assert maxdiff(rgb.cpu().reshape(-1, 3), ref.fine.rgb.cpu().reshape(-1, 3)) <= 1e-5      # ray direction fp32 rounding only
    rgb_h, depth_h, ev = rend.frame_to_host_async(rgb, depth)
    ev.synchronize()
    assert torch.equal(rgb_h, rgb.cpu()) and torch.equal(depth_h, depth.cpu())


def test_encoder_half_channels_last_latents_are_consumed_without_repack():
    """N2: the ResNet trunk run in fp16 channels-last leaves latents that ARE the kernel's packed layout; they are
    handed to libpnr_hip by pointer (no pnr_pack_latents), and the render matches the fp32-trunk + repack route.
    (The trunk itself is unpinned against torchvision, which is not installed here: parity unpinned for N2.)"""
    import ctypes as C
    from hip_util import model_conf, build_renderer
    import golden_util as gu
    from pixel_nerf_multiscale_amd import PixelNeRFNet, util
    spec = dict(gu.CASES["full_ns3"]); spec.update(NS=2, Kf=0, Kfd=0, Kc=32)
    torch.manual_seed(0)
    net = PixelNeRFNet(model_conf(spec, "fp16"))
    for which, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)):
        mlp.load_state_dict({k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, which).items()})
    net = net.cuda().eval()
    W, H = spec["image"]
    images = torch.rand(1, 2, 3, H, W, device="cuda") * 2 - 1
    poses = torch.stack([util.pose_spherical(30.0 * v, -20.0, spec["radius"]) for v in range(2)])[None].cuda()
    rays = util.gen_rays(util.pose_spherical(75.0, -25.0, spec["radius"])[None], W, H, torch.tensor(spec["focal"]),
                         spec["z_near"], spec["z_far"]).reshape(1, -1, 8)[:, ::3].contiguous().cuda()
    rend = build_renderer(spec)
    rend.forced_seed = 3
    with torch.no_grad():
        net.encode(images, poses, torch.tensor(spec["focal"]))
    lat32 = net.encoder.latent.clone()
    assert lat32.shape == (2, 256, H // 16, W // 16)
    ref = rend(net, rays).coarse.rgb.cpu()                         # fp32 trunk -> pnr_pack_latents -> fp16 kernel
    net.encoder.half_dtype = torch.float16
    with torch.no_grad():
        net.encode(images, poses, torch.tensor(spec["focal"]))
    m16 = net.encoder.level_maps16(torch.float16)
    assert m16 is not None and m16[0].dtype == torch.float16 and m16[0].is_contiguous(memory_format=torch.channels_last)
    rel = float((m16[0].float() - lat32).abs().max() / lat32.abs().max())
    assert rel < 2e-2
    v, keep = net.views_struct("fp16")
    assert v.latent_packed[0] == m16[0].data_ptr()                 # zero copy
    out = rend(net, rays).coarse.rgb.cpu()
    assert _psnr(out, ref) >= 45.0


@pytest.mark.parametrize("Kc,Kf,Kfd", [(1, 0, 0), (37, 0, 0), (5, 3, 1), (130, 70, 30), (300, 212, 100)])
def test_stage_kernels_ragged_sample_counts(Kc, Kf, Kfd):
    """Sample counts that are not multiples of the wave size / not powers of two, one sample per ray, and a merged
    count above 256 (bitonic padding to 512): stage kernels vs the oracle with identical explicit noise."""
    from pixel_nerf_multiscale_amd import NeRFRenderer
    g = torch.Generator().manual_seed(Kc * 7 + Kf)
    N = 67
    rays = torch.zeros(N, 8)
    rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    rays[:, 6] = 0.5 + torch.rand(N, generator=g)
    rays[:, 7] = rays[:, 6] + 1.0 + torch.rand(N, generator=g)
    n_imp = Kf - Kfd
    noise = dict(noise_c=torch.rand(N, Kc, generator=g))
    if n_imp > 0:
        noise.update(u=torch.rand(N, n_imp, generator=g), r=torch.rand(N, n_imp, generator=g))
    if Kfd > 0:
        noise.update(g=torch.randn(N, Kfd, generator=g))
    rend = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=0.05, white_bkgd=False).cuda()
    rend.fixed_noise = {k: v.cuda() for k, v in noise.items()}
    zc_ref = orc.sample_coarse(rays, Kc, False, noise["noise_c"])
    zc = rend.sample_coarse(rays.cuda())
    assert maxdiff(zc.cpu(), zc_ref) <= 1e-5
    out = torch.rand(N, Kc, 4, generator=g)
    out[..., 3] *= 25.0
    w_ref, rgb_ref, d_ref = orc.composite(rays, zc_ref, out, False)
    w, rgb, d = rend._composite_native(rays.cuda(), zc_ref.cuda().contiguous(), out.cuda())
    assert maxdiff(w.cpu(), w_ref) <= 5e-6 and maxdiff(rgb.cpu(), rgb_ref) <= 2e-5 and maxdiff(d.cpu(), d_ref) <= 1e-4
    if Kf > 0:
        samps = [zc_ref]
        if n_imp > 0:
            samps.append(orc.sample_fine(rays, w_ref, Kc, False, noise["u"], noise["r"]))
        if Kfd > 0:
            samps.append(orc.sample_fine_depth(rays, d_ref, 0.05, noise["g"]))
        zf_ref = torch.sort(torch.cat(samps, -1), -1)[0]
        rend.last_seed = 0
        zf = rend.sample_fine_sorted(rays.cuda(), zc_ref.cuda(), w_ref.cuda(), d_ref.cuda())
        # a cdf entry within 1 ulp of a draw may land in the neighbouring bin (documented): allow a handful of samples
        bad = (zf.cpu() - zf_ref).abs() > 1e-4
        assert int(bad.sum()) <= max(2, int(2e-4 * zf_ref.numel()))


def test_c_abi_error_codes():
    """Reference convention is a Python assert (nerf.py:269, resnetfc.py:190); the C ABI reports codes, the Python
    layer raises."""
    import ctypes as C
    from hip_util import setup
    from pixel_nerf_multiscale_amd import _native as N
    fx, spec, net, rend = setup("full_ns1", precision="bf16")
    rays = _dev(fx["rays"]).reshape(-1, 8)
    prm = net.params_struct(rend, "bf16")
    m, k1 = net.mlp_struct(net.mlp_coarse, "bf16")
    v, k2 = net.views_struct("bf16")
    o = N.pnr_outputs()
    rgb = torch.empty(16, 3, device="cuda"); dep = torch.empty(16, device="cuda")
    o.coarse_rgb, o.coarse_depth, o.fine_rgb, o.fine_depth = N.ptr(rgb), N.ptr(dep), N.ptr(rgb), N.ptr(dep)
    nbytes = N.lib.pnr_workspace_bytes(C.byref(prm), C.byref(m), C.byref(v), 16)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    s = N.current_stream(rays.device)
    call = lambda p=prm, mm=m, vv=v, nb=nbytes, n=16, per=16: N.lib.pnr_render(
        C.byref(p), C.byref(mm), None, C.byref(vv), N.ptr(rays), n, per, None, 1, 0, C.byref(o), ws.data_ptr(), nb, s)
    assert call() == 0
    assert call(nb=nbytes // 2) == -4                                   # workspace too small
    assert call(n=16, per=5) == -2                                      # rays not divisible into objects
    p2 = N.pnr_params.from_buffer_copy(prm); p2.precision = N.PNR_F16
    assert call(p=p2) == -6                                             # packed for bf16, asked for fp16
    m2 = N.pnr_mlp.from_buffer_copy(m); m2.packed = None
    assert call(mm=m2) == -6                                            # packed stream missing
    m3 = N.pnr_mlp.from_buffer_copy(m); m3.d_in = 41
    assert call(mm=m3) == -2
    m4 = N.pnr_mlp.from_buffer_copy(m); m4.combine_type = 7
    assert call(mm=m4) == -3
    p5 = N.pnr_params.from_buffer_copy(prm); p5.n_fine_depth = 99
    assert call(p=p5) == -2
    assert N.lib.pnr_composite(N.ptr(rays), rays.data_ptr() + 4, rays.data_ptr() + 4, 2, 2, 0, None, N.ptr(rgb), N.ptr(dep), s) == -5
    torch.cuda.synchronize()


@pytest.mark.parametrize("prec,floor", [("bf16", BF16_FLOOR_DB), ("fp16", FP16_FLOOR_DB)])
def test_mfma_frame_psnr_dtu_cfg4(prec, floor):
    """BASELINE cfg4 shape at frame level: 3 views, 19x25 latent (475 texels: the general gather + lin_z stream at
    NS=3, the view park / reduce path), 128 samples in disparity over z in [0.1, 5], black background; 4099 rays of the
    400x300 frame (a tail tile), fused kernel vs the fp32 HIP path (pinned to the reference by full_dtu_ns3) with
    identical in-kernel noise."""
    from hip_util import build_net, build_renderer
    spec = dict(gu.CASES["full_dtu_ns3"])
    poses = np.stack([gu.pose_spherical(30.0 * v, -20.0, spec["radius"]) for v in range(3)])[None]
    g = torch.Generator().manual_seed(4)
    W, H = spec["image"]
    tgt = gu.pose_spherical(75.0, -25.0, spec["radius"])
    rays = torch.from_numpy(gu.pinhole_rays(tgt, W, H, spec["focal"], spec["z_near"], spec["z_far"],
                                            torch.randperm(W * H, generator=g)[:4099].numpy()))[None].cuda()
    outs = {}
    for p in ("fp32", prec):
        net = build_net(spec, poses, "cuda", p)
        rend = build_renderer(spec)
        rend.forced_seed = 17
        o = rend(net, rays, want_weights=True)
        outs[p] = (o.coarse.rgb.cpu(), o.coarse.weights.cpu())
    assert not torch.isnan(outs[prec][0]).any()
    assert _psnr(outs[prec][0], outs["fp32"][0]) >= floor
    assert float(outs[prec][1].sum(-1).max()) <= 1.0 + 1e-3


def test_sharded_renderer_world1_equals_forward_on_the_hip_renderer():
    """a16: the HIP renderer under ShardedRenderer / bind_parallel (world = 1: no collective) returns exactly
    forward()'s pixels for the same seed; the sharded form keys the noise by the global ray index."""
    from hip_util import setup
    from pixel_nerf_multiscale_amd.parallel import ShardedRenderer, frame_seed
    fx, spec, net, rend = setup("tiny_ns2_codeview")
    rend.fixed_noise = None
    rays = _dev(fx["rays"])
    sr = ShardedRenderer.for_model(rend, net, base_seed=21)
    rgb, depth = sr(rays)
    rend.forced_seed = frame_seed(21, 0)
    ref = rend(net, rays)
    rend.forced_seed = None
    assert torch.equal(rgb, ref.fine.rgb) and torch.equal(depth, ref.fine.depth)
    with pytest.warns(UserWarning):
        wrapped = rend.bind_parallel(net, gpus=[0, 1], simple_output=True)      # no process group: one device
    rend.forced_seed = frame_seed(21, 0)
    rgb2, depth2 = wrapped(rays)
    rend.forced_seed = None
    assert torch.equal(rgb2, ref.fine.rgb)


def _staged_render(net, rend, rays):
    """The passes of NeRFRenderer.forward as SEPARATE launches through the stage entry points (pnr_sample_coarse,
    pnr_point_mlp on (rays, z), pnr_composite, pnr_sample_fine): what pnr_render did before its launches were fused."""
    import ctypes as C
    from pixel_nerf_multiscale_amd import _native as N
    SB, B, _ = rays.shape
    dev = rays.device
    r = N.f32c(rays).reshape(-1, 8)
    n = SB * B
    fine = rend.using_fine
    prec = net.resolved_precision(net.mlp_coarse, net.mlp_fine if fine else None)
    prm = net.params_struct(rend, prec)
    v, keep_v = net.views_struct(prec)

    def one_pass(mlp_mod, z):
        K = z.shape[1]
        m, keep_m = net.mlp_struct(mlp_mod, prec, v)
        out = torch.empty(n * K, 4, device=dev)
        ws = torch.empty(N.lib.pnr_workspace_bytes(C.byref(prm), C.byref(m), C.byref(v), n), dtype=torch.uint8, device=dev)
        N.check(N.lib.pnr_point_mlp(C.byref(prm), C.byref(m), C.byref(v), N.ptr(r), N.ptr(z), K, None, None, n * K, B * K,
                                    N.ptr(out), ws.data_ptr(), ws.numel(), N.current_stream(dev)), "pnr_point_mlp")
        return rend._composite_native(r, z, out)

    zc = rend.sample_coarse(r)
    res = {"coarse": one_pass(net.mlp_coarse, zc) + (zc,)}
    if fine:
        zf = rend.sample_fine_sorted(r, zc, res["coarse"][0], res["coarse"][2])
        res["fine"] = one_pass(net.mlp_fine if net.mlp_fine is not None else net.mlp_coarse, zf) + (zf,)
    return res


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("name", FULL)
def test_fused_render_launch_is_bit_identical_to_the_staged_launches(name, prec):
    """pnr_render runs a pass as ONE launch of the MFMA kernel (coarse positions generated in the tile
    prologue, the network, compositing of every finished ray by the workgroup that evaluated it).  Same arithmetic in the
    same order as the stage kernels, so pixels, depths, weights and sample positions are BIT-identical to the staged
    launches — on the reference fixtures (explicit noise tensors) and with in-kernel Philox noise on a ragged ray count."""
    from hip_util import setup
    fx, spec, net, rend = setup(name, precision=prec)
    rend.keep_samples = True
    rays = _dev(fx["rays"])
    for mode in ("fixture noise", "kernel rng"):
        if mode == "kernel rng":
            rend.fixed_noise = None
            rend.forced_seed = 4242
            rays = rays[:, : rays.shape[1] - 3]          # ragged: the last workgroup's range ends inside a tile
        fused = rend(net, rays, want_weights=True)
        staged = _staged_render(net, rend, rays)
        for lvl, (w, rgb, depth, z) in staged.items():
            SB, B = rays.shape[:2]
            assert torch.equal(fused[lvl].z.reshape(SB * B, -1), z), (mode, lvl, "z")
            assert torch.equal(fused[lvl].weights.reshape(SB * B, -1), w), (mode, lvl, "weights")
            assert torch.equal(fused[lvl].rgb.reshape(SB * B, 3), rgb), (mode, lvl, "rgb")
            assert torch.equal(fused[lvl].depth.reshape(SB * B), depth), (mode, lvl, "depth")


@pytest.mark.parametrize("name,prec", [("full_ns1", "bf16"), ("full_ns3", "fp16"), ("full_dtu_ns3", "bf16"), ("tiny_ns2_codeview", "fp32")])
def test_rays_generated_inside_the_render_launch_are_bit_identical(name, prec):
    """N1: NeRFRenderer.render_image -> pnr_render_camera: the MFMA kernel forms each point's ray from (c2w, intrinsics, pixel
    index) in its tile prologue; no ray tensor exists.  Bit-identical to pnr_gen_rays + forward on those rays (same in-kernel
    noise), for both principal-point conventions and a non-square image; the fp32 path materialises the rays in the workspace."""
    from hip_util import setup
    from pixel_nerf_multiscale_amd import util
    fx, spec, net, rend = setup(name, precision=prec)
    rend.fixed_noise = None
    rend.forced_seed = 31
    pose = util.pose_spherical(40.0, -30.0, spec["radius"])
    for (W, H, c) in ((24, 17, None), (19, 23, (9.25, 12.5))):
        f = (spec["focal"], spec["focal"] * 1.1)
        rgb, depth = rend.render_image(net, pose, W, H, f, spec["z_near"], spec["z_far"], c=c)
        rays = util.gen_rays_device(pose, W, H, f, spec["z_near"], spec["z_far"], c=c, device="cuda")
        ref = rend(net, rays[None])
        lvl = ref.fine if rend.using_fine else ref.coarse
        assert torch.equal(rgb.reshape(-1, 3), lvl.rgb.reshape(-1, 3)), (W, H)
        assert torch.equal(depth.reshape(-1), lvl.depth.reshape(-1)), (W, H)


@pytest.mark.parametrize("name", ["tiny_ns1", "tiny_sb2_ns2", "tiny_max_combine", "full_ns3", "tiny_multiscale_ns2"])
def test_resnetfc_forward_and_encoder_index_are_native_stage_calls(name):
    """ResnetFC.forward (resnetfc.py:173-236) and SpatialEncoder.index (encoder.py:138-205) as module calls backed by
    pnr_resnetfc_forward / pnr_index_latent, against the oracle's restatement of the same two functions: assembled rows
    with the (views, points) reduction, a flat batch without one, a row count that crosses the 16384-row chunk, and the
    lookup with per-view and broadcast uv incl. off-image points."""
    import oracle_util as ou
    from hip_util import setup
    from oracle import pixelnerf_oracle as orc
    fx, spec, net, rend = setup(name)
    g = torch.Generator().manual_seed(5)
    mlp = net.mlp_coarse
    sd = {k: v.detach().cpu() for k, v in mlp.state_dict().items()}
    SB, NS = spec["SB"], spec["NS"]
    E = mlp.d_latent + mlp.d_in
    for P in (7, 16400 if name == "tiny_ns1" else 33):
        zx = torch.randn(SB * NS * P, E, generator=g)
        ref = orc.resnetfc(sd, zx, mlp.d_latent, NS, P, n_blocks=mlp.n_blocks, combine_layer=mlp.combine_layer,
                           combine_type=mlp.combine_type)
        out = mlp(zx.cuda(), combine_inner_dims=(NS, P))
        assert tuple(out.shape) == ((SB, P, mlp.d_out) if NS > 1 else (SB * NS * P, mlp.d_out))
        scale = max(1.0, float(ref.abs().max()))
        assert maxdiff(out.reshape(-1, mlp.d_out).cpu(), ref) <= 2e-5 * scale, P
    flat = torch.randn(3, 5, E, generator=g)             # no reduction: leading dims are kept
    ref = orc.resnetfc(sd, flat.reshape(-1, E), mlp.d_latent, 1, 15, n_blocks=mlp.n_blocks, combine_layer=mlp.combine_layer,
                       combine_type=mlp.combine_type)
    out = mlp(flat.cuda())
    assert tuple(out.shape) == (3, 5, mlp.d_out) and maxdiff(out.reshape(-1, mlp.d_out).cpu(), ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # the lookup
    maps = [m.cpu() for m in net.encoder.level_maps()]
    nv = maps[0].shape[0]
    W, H = spec["image"]
    uv = (torch.rand(nv, 41, 2, generator=g) * 1.4 - 0.2) * torch.tensor([float(maps[-1].shape[3]), float(maps[-1].shape[2])])
    for q in (uv, uv[:1]):
        ref = orc.index_latent(q, maps)
        out = net.encoder.index(q.cuda())
        assert tuple(out.shape) == tuple(ref.shape)
        assert maxdiff(out.cpu(), ref) <= 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("Kc,Kf,Kfd,n_rays", [(150, 100, 30, 5), (1, 0, 0, 131), (37, 11, 11, 131), (128, 64, 0, 3), (300, 212, 100, 2),
                                               (24, 16, 8, 2300)])       # > 2048 rays: the resampling is its own launch
@pytest.mark.parametrize("name,prec", [("full_ns1", "bf16"), ("full_ns3", "fp16")])
def test_fused_render_launch_ragged_sample_counts(name, prec, Kc, Kf, Kfd, n_rays):
    """The fused launch against the staged launches where rays do not line up with the 128-point tiles: rays longer than a
    tile (a ray is finished — and composited — several tiles after it started), one sample per ray, sample counts that are
    not multiples of anything, fewer rays than workgroups, and both resampling routes (inside the fine launch up to 2048
    rays, a separate launch above); in-kernel noise.  Bit-identical pixels, depths, weights, positions."""
    from hip_util import setup
    fx, spec, net, rend = setup(name, precision=prec)
    rend.fixed_noise = None
    rend.forced_seed = 777
    rend.keep_samples = True
    rend.n_coarse, rend.n_fine, rend.n_fine_depth = Kc, Kf, Kfd
    rend.using_fine = Kf > 0
    W, H = spec["image"]
    g = torch.Generator().manual_seed(Kc * 7 + n_rays)
    tgt = gu.pose_spherical(50.0, -20.0, spec["radius"])
    pix = torch.randperm(W * H, generator=g)[:n_rays].numpy()
    rays = torch.from_numpy(gu.pinhole_rays(tgt, W, H, spec["focal"], spec["z_near"], spec["z_far"], pix))[None].cuda()
    fused = rend(net, rays, want_weights=True)
    staged = _staged_render(net, rend, rays)
    for lvl, (w, rgb, depth, z) in staged.items():
        assert torch.equal(fused[lvl].z.reshape(n_rays, -1), z), (lvl, "z")
        assert torch.equal(fused[lvl].weights.reshape(n_rays, -1), w), (lvl, "weights")
        assert torch.equal(fused[lvl].rgb.reshape(n_rays, 3), rgb), (lvl, "rgb")
        assert torch.equal(fused[lvl].depth.reshape(n_rays), depth), (lvl, "depth")


@pytest.mark.parametrize("NS,SB,N", [(2, 2, 333), (1, 3, 50), (3, 2, 129)])
def test_fused_render_launch_several_objects(NS, SB, N):
    """Several objects per call (SB > 1): a workgroup's ray range may cross from one object to the next inside a tile, every
    point picking its own object's cameras and latents.  Fused launch == staged launches, bit for bit (bf16)."""
    from hip_util import build_net, build_renderer
    spec = dict(gu.CASES["full_ns1"]); spec.update(NS=NS, SB=SB, N=N, seed=90 + NS + SB)
    rays_np, poses = gu.make_inputs(spec)
    rays = torch.from_numpy(rays_np).cuda()
    net = build_net(spec, poses, "cuda", "bf16")
    rend = build_renderer(spec)
    rend.forced_seed = 17
    rend.keep_samples = True
    fused = rend(net, rays, want_weights=True)
    staged = _staged_render(net, rend, rays)
    for lvl, (w, rgb, depth, z) in staged.items():
        assert torch.equal(fused[lvl].z.reshape(SB * N, -1), z), (lvl, "z")
        assert torch.equal(fused[lvl].weights.reshape(SB * N, -1), w), (lvl, "weights")
        assert torch.equal(fused[lvl].rgb.reshape(SB * N, 3), rgb), (lvl, "rgb")
        assert torch.equal(fused[lvl].depth.reshape(SB * N), depth), (lvl, "depth")


@pytest.mark.parametrize("prec,floor", [("bf16", 50.0), ("fp16", 65.0)])
@pytest.mark.parametrize("NS,SB,N,lat", [(1, 2, 700, (256, 8, 8)), (2, 3, 257, (256, 6, 6)), (1, 5, 40, (256, 8, 8)),
                                          (2, 2, 300, "multiscale")])
def test_projected_stream_several_objects(prec, floor, NS, SB, N, lat):
    """Projected streams for SB > 1: one stream per object (each carries W_z . Lat of ITS views); workgroups are assigned per
    object, so a tile never mixes objects.  Against the general (gather + lin_z) stream and the fp32 path on the same rays;
    explicit points through PixelNeRFNet.forward as well (the plain launch's per-object tile assignment)."""
    from hip_util import build_net, build_renderer
    cv = False
    if lat == "multiscale":
        lats, cv = [(64, 32, 32), (64, 32, 32), (128, 16, 16), (256, 8, 8)], True
    else:
        lats = [lat]
    spec = dict(gu.CASES["full_ns1"]); spec.update(NS=NS, SB=SB, N=N, Kc=40, Kf=24, Kfd=8, lat=lats, seed=60 + NS + SB,
                                                    use_code_viewdirs=cv)
    rays_np, poses = gu.make_inputs(spec)
    rays = torch.from_numpy(rays_np).cuda()
    outs, pts = {}, {}
    g = torch.Generator().manual_seed(9)
    xyz = ((torch.rand(SB, 333, 3, generator=g) - 0.5) * 1.2).cuda()
    vd = torch.nn.functional.normalize(torch.randn(SB, 333, 3, generator=g), dim=-1).cuda()
    for tag, p, proj in (("fp32", "fp32", False), ("proj", prec, True), ("gen", prec, False)):
        net = build_net(spec, poses, "cuda", p)
        net.project_latent = proj
        if p != "fp32":
            v, _ = net.views_struct(p)
            m, _ = net.mlp_struct(net.mlp_coarse, p, v)
            assert (m.packed_texels > 0) == proj and m.packed_objs == (SB if proj else 0)
        rend = build_renderer(spec)
        rend.forced_seed = 7
        outs[tag] = rend(net, rays).coarse.rgb.cpu()
        pts[tag] = net(xyz, coarse=True, viewdirs=vd).cpu()
    assert _psnr(outs["proj"], outs["fp32"]) >= floor and _psnr(outs["gen"], outs["fp32"]) >= floor
    assert _psnr(outs["proj"], outs["gen"]) >= floor
    assert _psnr(pts["proj"][..., :3], pts["fp32"][..., :3]) >= floor
    assert _psnr(pts["proj"][..., :3], pts["gen"][..., :3]) >= floor


def test_full_dtu_frame_at_baseline_size():
    """BASELINE cfg 4 at its full size — one 400 x 300 frame, 3 source views, 128 samples per ray, disparity sampling, black
    background (120 000 rays, 15.4 M points in one launch) — through the size-independent properties the domain offers, plus
    the identity of the two ray sources at that size: rays from the tensor vs rays formed in the kernel from the camera, and
    the first / second half of the frame rendered as separate calls with ray_index_base (the sharding identity)."""
    from hip_util import build_net, build_renderer
    from pixel_nerf_multiscale_amd import util
    spec = dict(gu.CASES["full_dtu_ns3"])
    poses = np.stack([gu.pose_spherical(30.0 * v, -20.0, spec["radius"]) for v in range(spec["NS"])])[None]
    net = build_net(spec, poses, "cuda", "bf16")
    rend = build_renderer(spec)
    rend.forced_seed = 123
    W, H = spec["image"]
    assert (W, H, spec["Kc"], spec["NS"]) == (400, 300, 128, 3)
    pose = util.pose_spherical(75.0, -25.0, spec["radius"])
    rgb, depth = rend.render_image(net, pose, W, H, spec["focal"], spec["z_near"], spec["z_far"])
    assert rgb.shape == (H, W, 3) and bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 + 1e-5            # sum w c, c in [0,1], sum w <= 1, black background
    assert float(depth.min()) >= 0.0 and float(depth.max()) <= spec["z_far"] + 1e-3
    rays = util.gen_rays_device(pose, W, H, spec["focal"], spec["z_near"], spec["z_far"], device="cuda")
    ref = rend(net, rays[None], want_weights=True)
    assert torch.equal(rgb.reshape(-1, 3), ref.coarse.rgb[0]) and torch.equal(depth.reshape(-1), ref.coarse.depth[0])
    w = ref.coarse.weights[0]
    assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
    half = rays.shape[0] // 2
    parts = []
    for lo, hi in ((0, half), (half, rays.shape[0])):
        rend.ray_index_base = lo
        parts.append(rend(net, rays[None, lo:hi]).coarse.rgb[0])
    rend.ray_index_base = 0
    assert torch.equal(torch.cat(parts), ref.coarse.rgb[0])


# ----------------------------------------------------------------------------- the fused kernel's shape space
# make_layout (csrc/point_mfma.hip) accepts n_blocks 1..8, any combine_layer, d_latent in {256, 512, 768, 1024}: every
# corner below either renders within the stated bound of the ORACLE (reference resnetfc.py:128-158,203-234 restated for
# any n_blocks / combine_layer) or is refused with PNR_E_UNSUPPORTED before anything is launched.
_SHAPES = [
    # n_blocks, combine_layer, NS, latent levels, project_latent
    (5, 0, 1, [(256, 8, 8)], True),                                          # no lin_z block at all (nb1 == 0)
    (5, 5, 1, [(256, 8, 8)], True),                                          # every block has lin_z, none behind it (projected)
    (5, 5, 1, [(256, 8, 8)], False),                                         # the same on the general stream
    (5, 1000, 1, [(256, 19, 25)], True),                                     # the class default combine_layer, map too large to project
    (3, 1, 2, [(256, 8, 8)], True),
    (8, 3, 3, [(256, 6, 6)], True),
    (8, 3, 3, [(256, 6, 6)], False),
    (1, 1, 1, [(256, 8, 8)], True),
    (2, 1, 2, [(256, 8, 8)], False),
    (4, 2, 1, [(64, 16, 16), (64, 16, 16), (128, 8, 8), (256, 4, 4)], True),   # d_latent 512, last level projected
    (5, 3, 2, [(512, 8, 8), (256, 4, 4)], True),                             # d_latent 768: two gathered groups + projection
    (5, 3, 2, [(256, 8, 8), (256, 8, 8), (256, 19, 25)], True),              # d_latent 768, three gathered groups
    (3, 2, 1, [(256, 8, 8), (256, 8, 8), (256, 8, 8), (256, 8, 8)], False),  # d_latent 1024
]


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("n_blocks,combine_layer,NS,lat,proj", _SHAPES)
def test_fused_kernel_shape_space(n_blocks, combine_layer, NS, lat, proj, prec):
    from hip_util import build_net, build_renderer
    from oracle import pixelnerf_oracle as orc
    import golden_util as gu
    spec = dict(gu.CASES["full_ns1"])
    spec.update(n_blocks=n_blocks, combine_layer=combine_layer, NS=NS, lat=lat, N=150, Kc=16, Kf=8, Kfd=4,
                seed=300 + 7 * n_blocks + combine_layer % 11 + NS, use_code_viewdirs=len(lat) == 4)
    rays_np, poses = gu.make_inputs(spec)
    g = torch.Generator().manual_seed(spec["seed"])
    n, n_imp = spec["N"], spec["Kf"] - spec["Kfd"]
    noise = dict(noise_c=torch.rand(n, spec["Kc"], generator=g), u=torch.rand(n, n_imp, generator=g),
                 r=torch.rand(n, n_imp, generator=g), g=torch.randn(n, spec["Kfd"], generator=g))
    W, H = spec["image"]
    cam = orc.encode_cameras(torch.from_numpy(poses), spec["focal"], None, W, H)
    sd = {w: {k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, w).items()} for w in ("coarse", "fine")}
    with torch.no_grad():
        ref = orc.render(sd["coarse"], sd["fine"], cam, [torch.from_numpy(x) for x in gu.make_latents(spec)],
                         torch.from_numpy(rays_np), NS, spec["Kc"], spec["Kf"], spec["Kfd"], spec["depth_std"],
                         spec["white_bkgd"], spec["lindisp"], noise, use_code_viewdirs=spec["use_code_viewdirs"],
                         n_blocks=n_blocks, combine_layer=combine_layer, combine_type=spec["combine_type"])
    rays = torch.from_numpy(rays_np).cuda()
    nets, outs = {}, {}
    for p in ("fp32", prec):
        net = build_net(spec, poses, "cuda", p)
        net.project_latent = proj
        assert net.resolved_precision(net.mlp_coarse, net.mlp_fine) == p
        rend = build_renderer(spec)
        rend.fixed_noise = {k: v.cuda() for k, v in noise.items()}
        outs[p] = rend(net, rays, want_weights=True)
        nets[p] = (net, rend)
    # the fp32 HIP path on this shape is pinned to the oracle like the fixtures pin it on the shipped shape
    for lvl in ("coarse", "fine"):
        assert maxdiff(outs["fp32"][lvl].rgb.cpu(), ref[lvl]["rgb"]) <= 1e-4, lvl
        # per-sample weights: sigma is O(80) in these synthetic networks, so alpha = 1 - exp(-delta sigma) carries the fp32
        # summation-order difference of the last layer at a few 1e-4 (measured 1.4e-4); a flipped bin would show as ~1e-1
        assert maxdiff(outs["fp32"][lvl].weights.cpu(), ref[lvl]["weights"]) <= 4e-4, lvl
    assert _psnr(outs[prec].coarse.rgb.cpu(), ref["coarse"]["rgb"]) >= FLOOR_DB[prec]
    assert not torch.isnan(outs[prec].fine.rgb).any()
    # the fine pass against the (now pinned) fp32 path on 1500 rays with in-kernel noise: the precision bound at the fp32
    # path's sample positions, the end-to-end floor on a sample large enough that ONE flipped bin does not decide it
    # (on the 150 rays above a single flip moved fp16 to 49.6 dB on one shape and bf16 to 39.97 dB on another)
    # ... at the reference's sample schedule (64 + 32, conf/default.conf:50-53), where the statistic means something
    spec2 = dict(spec); spec2.update(N=1500)
    rays2 = torch.from_numpy(gu.make_inputs(spec2)[0]).cuda()
    K = 64 + 32
    o2, inj = {}, {}
    for p in ("fp32", prec):
        net, rend = nets[p]
        rend.n_coarse, rend.n_fine, rend.n_fine_depth = 64, 32, 16
        rend.fixed_noise, rend.forced_seed, rend.keep_samples = None, 11, True
        o2[p] = rend(net, rays2)
        z32 = o2["fp32"].fine.z
        xyz = (rays2[:, :, None, :3] + z32[..., None] * rays2[:, :, None, 3:6]).reshape(1, -1, 3).contiguous()
        vd = rays2[:, :, None, 3:6].expand(-1, -1, K, -1).reshape(1, -1, 3).contiguous()
        pts = net(xyz, coarse=False, viewdirs=vd).reshape(-1, K, 4).contiguous()
        inj[p] = rend._composite_native(rays2.reshape(-1, 8), z32.reshape(-1, K).contiguous(), pts)[1].cpu()
    assert _psnr(inj["fp32"], o2["fp32"].fine.rgb.cpu().reshape(-1, 3)) >= 90.0
    assert _psnr(o2[prec].coarse.rgb.cpu(), o2["fp32"].coarse.rgb.cpu()) >= FLOOR_DB[prec]
    assert _psnr(inj[prec], inj["fp32"]) >= FLOOR_DB[prec], "fine pass at the fp32 sample positions"
    assert _psnr(o2[prec].fine.rgb.cpu(), o2["fp32"].fine.rgb.cpu()) >= FINE_E2E_FLOOR_DB[prec], "fine, end to end"


@pytest.mark.parametrize("n_blocks,combine_layer,NS,lat", [
    (5, 0, 2, [(256, 8, 8)]),                       # view reduction in front of the first block: nothing per view to run
    (3, 5, 2, [(256, 8, 8)]),                       # reduction behind the last block: the reference returns NS x the rows
    (9, 3, 1, [(256, 8, 8)]),                       # more blocks than the bias table holds
])
def test_fused_kernel_refuses_what_it_does_not_run(n_blocks, combine_layer, NS, lat):
    from hip_util import build_net, build_renderer
    import golden_util as gu
    spec = dict(gu.CASES["full_ns1"])
    spec.update(n_blocks=n_blocks, combine_layer=combine_layer, NS=NS, lat=lat, N=8, Kf=0, Kfd=0)
    rays_np, poses = gu.make_inputs(spec)
    net = build_net(spec, poses, "cuda", "fp16")
    with pytest.raises((ValueError, RuntimeError)):
        build_renderer(spec)(net, torch.from_numpy(rays_np).cuda())
    torch.cuda.synchronize()


def test_two_streams_through_one_net_do_not_share_a_workspace():
    """pnr.h: every call is re-entrant, the caller owns the workspace — so PixelNeRFNet hands out one workspace per
    (device, stream, thread).  Two streams rendering different multi-view batches through ONE net at the same time (the
    multi-view workspace holds the parked per-view streams: sharing it would corrupt both) equal the sequential renders."""
    from hip_util import build_net, build_renderer
    import golden_util as gu
    spec = dict(gu.CASES["full_ns3"]); spec.update(N=3000, Kc=64, Kf=0, Kfd=0, seed=55)
    rays_np, poses = gu.make_inputs(spec)
    rays = torch.from_numpy(rays_np).cuda()
    a, b = rays[:, :1500].contiguous(), rays[:, 1500:].contiguous()
    net = build_net(spec, poses, "cuda", "fp16")
    rend = build_renderer(spec)
    rend.forced_seed = 21
    ref_a, ref_b = rend(net, a).coarse.rgb.clone(), rend(net, b).coarse.rgb.clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(3):                       # several rounds in flight on both streams
        with torch.cuda.stream(s1):
            o1 = rend(net, a).coarse.rgb
        with torch.cuda.stream(s2):
            o2 = rend(net, b).coarse.rgb
        outs.append((o1, o2))
    torch.cuda.synchronize()
    assert len(net._ws) >= 3                 # the default stream's and the two side streams'
    for o1, o2 in outs:
        assert torch.equal(o1, ref_a) and torch.equal(o2, ref_b)
