"""N4 (SURVEY §8 f): the renderer under autograd — taped fp32 forward + explicit HIP backward kernels
(csrc/train_f32.hip) through the C ABI — against the REFERENCE's gradients (tests/golden/*_grad.npz, written
by tools/gen_golden_grad.py) and against the oracle under torch autograd.  Tolerance: fp32 arithmetic with a
different summation order (tiled GEMMs, atomics): 5e-4 of each tensor's scale."""
import numpy as np
import pytest
import torch

import golden_util as gu
import hip_util as hu
from test_oracle_grad import compare_grads, oracle_grads

pytestmark = pytest.mark.gpu
RTOL = 5e-4


def hip_grads(name, fx=None, train_precision="fp32"):
    fx, spec, net, rend = hu.setup(name)
    net.train()
    net.train_precision = train_precision
    maps = [torch.from_numpy(x).cuda().requires_grad_(True) for x in gu.make_latents(spec)]
    net.encoder.set_latents(maps)
    leaf_maps = net.encoder.level_maps()
    for m in leaf_maps:
        assert m.requires_grad
    rays = torch.from_numpy(fx["rays"]).cuda()
    out = rend(net, rays, want_weights=True)
    G = {k: torch.from_numpy(v).cuda() for k, v in gu.make_loss_weights(spec).items()}
    loss = 0.0
    for tag in ("coarse", "fine") if spec["Kf"] > 0 else ("coarse",):
        lvl = out[tag]
        loss = loss + (lvl.rgb * G[f"{tag}_rgb"]).sum() + (lvl.depth * G[f"{tag}_depth"]).sum() \
            + (lvl.weights * G[f"{tag}_weights"]).sum()
    loss.backward()
    grads = {}
    for which, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)):
        if mlp is not None:
            for k, p in mlp.named_parameters():
                if p.grad is not None:
                    grads[f"{which}.{k}"] = p.grad.cpu().numpy()
    for i, m in enumerate(maps):
        grads[f"latent.{i}"] = m.grad.cpu().numpy()
    return fx, out, float(loss.item()), grads


@pytest.mark.parametrize("name", gu.GRAD_CASES)
def test_gradients_match_reference(name):
    gfx = gu.load_grad_fixture(name)
    fx, out, loss, grads = hip_grads(name)
    # the taped forward is the fp32 path: same tolerance as the inference-precision parity tests
    assert np.abs(out.coarse.rgb.detach().cpu().numpy() - fx["coarse_rgb"]).max() <= 1e-4
    assert abs(loss - float(gfx["loss"])) <= 1e-3 * max(1.0, abs(float(gfx["loss"])))
    compare_grads(grads, gfx, RTOL, name, truth=lambda: oracle_grads(fx, torch.float64)[1])


@pytest.mark.parametrize("name", ["full_ns1", "full_ns3", "full_multiscale_ns2", "tiny_ns2_codeview", "tiny_max_combine"])
def test_bf16x3_gradients(name):
    """train_precision='bf16x3': every GEMM operand split hi + lo in bf16, three bf16 MFMA products per term, fp32
    accumulation — fp32-class results at several times the fp32-MFMA rate.  Against the fp32 path on the same inputs
    (coarse-pass cotangents: a 1e-5 change of the coarse outputs moves fine-pass samples across bins, which is a
    property of the renderer, not of the GEMMs): pixels within 1e-4, every gradient tensor within 1e-2 in l2 (measured
    1e-4 … 3e-3 on these steep synthetic nets; the plain bf16 mode sits at 5-8e-2)."""

    def run(prec):
        fx, spec, net, rend = hu.setup(name)
        net.train()
        net.train_precision = prec
        maps = [torch.from_numpy(x).cuda().requires_grad_(True) for x in gu.make_latents(spec)]
        net.encoder.set_latents(maps)
        out = rend(net, torch.from_numpy(fx["rays"]).cuda(), want_weights=True)
        G = {k: torch.from_numpy(v).cuda() for k, v in gu.make_loss_weights(spec).items()}
        loss = (out.coarse.rgb * G["coarse_rgb"]).sum() + (out.coarse.depth * G["coarse_depth"]).sum() \
            + (out.coarse.weights * G["coarse_weights"]).sum()
        loss.backward()
        grads = {"coarse." + k: p.grad.clone() for k, p in net.mlp_coarse.named_parameters()}
        grads.update({f"latent.{i}": m.grad.clone() for i, m in enumerate(maps)})
        return out.coarse.rgb.detach().clone(), grads

    rgb32, g32 = run("fp32")
    rgb3, g3 = run("bf16x3")
    assert float((rgb32 - rgb3).abs().max()) <= 1e-4
    for k in g32:
        a, b = g3[k].double().flatten(), g32[k].double().flatten()
        nb = float(b.norm())
        if nb == 0:
            continue
        rel = float((a - b).norm()) / nb
        assert rel <= 1e-2, (k, rel)


def test_composite_backward_vs_autograd():
    from oracle import pixelnerf_oracle as orc
    from pixel_nerf_multiscale_amd.render.autograd import Composite
    g = torch.Generator().manual_seed(5)
    for K, white in ((8, True), (64, False), (96, True), (160, False), (200, True)):
        B = 37
        z, _ = torch.sort(torch.rand(B, K, generator=g) * 1.5 + 1.25, dim=-1)
        rays = torch.cat([torch.randn(B, 6, generator=g), torch.full((B, 1), 1.25), torch.full((B, 1), 2.75)], -1)
        out = torch.cat([torch.rand(B, K, 3, generator=g), torch.relu(torch.randn(B, K, 1, generator=g) * 8 + 2)], -1)
        cot = [torch.randn(B, K, generator=g), torch.randn(B, 3, generator=g), torch.randn(B, generator=g)]
        zo, oo = z.clone().requires_grad_(True), out.clone().requires_grad_(True)
        w, rgb, d = orc.composite(rays, zo, oo, white)
        ((w * cot[0]).sum() + (rgb * cot[1]).sum() + (d * cot[2]).sum()).backward()
        zh, oh = z.cuda().requires_grad_(True), out.cuda().requires_grad_(True)
        w2, rgb2, d2 = Composite.apply(rays.cuda(), zh, oh, white)
        ((w2 * cot[0].cuda()).sum() + (rgb2 * cot[1].cuda()).sum() + (d2 * cot[2].cuda()).sum()).backward()
        for a, b, what in ((oh.grad, oo.grad, "d_out"), (zh.grad, zo.grad, "d_z")):
            scale = float(b.abs().max())
            assert float((a.cpu() - b).abs().max()) <= 2e-4 * scale + 1e-6, (K, white, what)
        # cotangents may be absent (unused outputs)
        zh2, oh2 = z.cuda().requires_grad_(True), out.cuda().requires_grad_(True)
        _, rgb3, _ = Composite.apply(rays.cuda(), zh2, oh2, white)
        (rgb3 * cot[1].cuda()).sum().backward()
        zo2, oo2 = z.clone().requires_grad_(True), out.clone().requires_grad_(True)
        _, rgb4, _ = orc.composite(rays, zo2, oo2, white)
        (rgb4 * cot[1]).sum().backward()
        assert float((oh2.grad.cpu() - oo2.grad).abs().max()) <= 2e-4 * float(oo2.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("name", ["tiny_ns2_codeview", "tiny_multiscale_ns2", "tiny_max_combine", "full_ns3"])
def test_point_forward_backward_explicit_points(name):
    """PixelNeRFNet.forward(xyz, viewdirs) in training mode: outputs and d(xyz), d(weights), d(latents) vs the
    oracle's point_forward under autograd."""
    from oracle import pixelnerf_oracle as orc
    from oracle_util import oracle_setup
    fx, spec, net, rend = hu.setup(name)
    net.train()
    spec, cam, lat, sd_c, sd_f = oracle_setup(fx)
    xyz = torch.from_numpy(fx["pts_xyz_coarse"]).clone()
    dirs = torch.from_numpy(fx["pts_dirs_coarse"]).clone()
    cot = torch.from_numpy(np.random.default_rng(3).standard_normal(fx["pts_out_coarse"].shape).astype(np.float32))
    # oracle
    xo = xyz.clone().requires_grad_(True)
    lo = [m.clone().requires_grad_(True) for m in lat]
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    oo = orc.point_forward(sdo, cam, lo, xo, dirs, spec["NS"], use_code_viewdirs=spec["use_code_viewdirs"],
                           n_blocks=spec["n_blocks"], combine_layer=spec["combine_layer"], combine_type=spec["combine_type"])
    (oo * cot).sum().backward()
    # HIP
    maps = [m.clone().cuda().requires_grad_(True) for m in lat]
    net.encoder.set_latents(maps)
    xh = xyz.cuda().requires_grad_(True)
    oh = net(xh, coarse=True, viewdirs=dirs.cuda())
    assert float((oh.detach().cpu() - oo.detach()).abs().max()) <= 1e-4 * max(1.0, float(oo.detach().abs().max()))
    (oh * cot.cuda()).sum().backward()

    def close(a, b, what):
        scale = max(float(b.abs().max()), float(b.norm()) / b.numel() ** 0.5)
        assert float((a.cpu() - b).abs().max()) <= RTOL * scale + 1e-7, (what, float((a.cpu() - b).abs().max()), scale)

    close(xh.grad, xo.grad, "d_xyz")
    for i, (a, b) in enumerate(zip(maps, lo)):
        close(a.grad, b.grad, f"d_latent{i}")
    for k, p in net.mlp_coarse.named_parameters():
        close(p.grad, sdo[k].grad, k)


def test_frozen_encoder_and_eval_mode_routing():
    """stop_encoder_grad detaches the latents (backup2:228-229); eval() / no_grad() keep the fused kernels."""
    fx, spec, net, rend = hu.setup("tiny_ns1")
    rays = torch.from_numpy(fx["rays"]).cuda()
    assert not net.wants_grad(rays)                       # eval mode
    net.train()
    assert net.wants_grad(rays)
    with torch.no_grad():
        assert not net.wants_grad(rays)
    net.stop_encoder_grad = True
    maps = [torch.from_numpy(x).cuda().requires_grad_(True) for x in gu.make_latents(spec)]
    net.encoder.set_latents(maps)
    out = rend(net, rays)
    out.fine.rgb.sum().backward()
    assert maps[0].grad is None
    assert net.mlp_fine.lin_out.weight.grad is not None and net.mlp_coarse.lin_out.weight.grad is not None


def test_backward_is_linear_at_training_batch_size():
    """Size-independent property at a realistic step (SB=2 x 128 rays x (64+32) samples, d_hidden 512, 2 views):
    grads(G1 + 2 G2) = grads(G1) + 2 grads(G2)."""
    spec = gu._case(seed=77, d_hidden=512, lat=[(256, 16, 16)], image=(128, 128), focal=131.25, NS=2, SB=2, N=128,
                    Kc=64, Kf=32, Kfd=16)
    rays_np, poses_np = gu.make_inputs(spec)
    net = hu.build_net(spec, poses_np).train()
    rend = hu.build_renderer(spec)
    rend.forced_seed = 1234
    rays = torch.from_numpy(rays_np).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    G1 = torch.randn(2, 128, 3, device="cuda", generator=g)
    G2 = torch.randn(2, 128, 3, device="cuda", generator=g)

    def run(G):
        net.zero_grad()
        out = rend(net, rays)
        (out.fine.rgb * G).sum().add((out.coarse.rgb * G).sum()).backward()
        return [p.grad.clone() for p in net.mlp_coarse.parameters()] + [p.grad.clone() for p in net.mlp_fine.parameters()]

    a, b, c = run(G1), run(G2), run(G1 + 2 * G2)
    for x, y, z in zip(a, b, c):
        ref = x + 2 * y
        assert float((z - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-6


def test_training_abi_error_codes_and_empty_batch():
    """Argument errors of the training entry points come back as PNR_E_* codes (no faults, no exceptions)."""
    import ctypes as C
    from pixel_nerf_multiscale_amd import _native as N
    fx, spec, net, rend = hu.setup("tiny_ns1")
    rays = torch.from_numpy(fx["rays"]).cuda().reshape(-1, 8)
    n, K = rays.shape[0], 8
    z = torch.rand(n, K, device="cuda").sort(dim=-1)[0] + 1.5
    prm = net.params_struct(None, "fp32")
    m, k1 = net.mlp_struct(net.mlp_coarse, "fp32")
    v, k2 = net.views_struct("fp32")
    s = N.current_stream(rays.device)
    P = n * K
    tb = N.lib.pnr_train_tape_bytes(C.byref(m), C.byref(v), P)
    wb = N.lib.pnr_train_bwd_workspace_bytes(C.byref(m), C.byref(v), P)
    assert tb > 0 and wb > 0
    tape = torch.empty(tb, dtype=torch.uint8, device="cuda")
    ws = torch.empty(wb, dtype=torch.uint8, device="cuda")
    out = torch.empty(P, 4, device="cuda")
    d_out = torch.randn(P, 4, device="cuda")
    g = N.pnr_mlp_grads()
    fwd = lambda tbytes=tb, npts=P, o=out: N.lib.pnr_point_mlp_train_fwd(
        C.byref(prm), C.byref(m), C.byref(v), N.ptr(rays), N.ptr(z), K, None, None, npts, npts, N.ptr(o) if o is not None else None,
        tape.data_ptr(), tbytes, s)
    assert fwd() == 0
    assert fwd(tbytes=tb // 2) == -4                       # tape too small
    assert fwd(npts=P - 1) == -2                           # not a multiple of K
    assert fwd(o=None) == -1
    bwd = lambda wbytes=wb, dxyz=None, grads=g: N.lib.pnr_point_mlp_bwd(
        C.byref(prm), C.byref(m), C.byref(v), N.ptr(rays), N.ptr(z), K, None, None, P, P, N.ptr(out), N.ptr(d_out),
        tape.data_ptr(), tb, C.byref(grads) if grads is not None else None, None, dxyz, None, ws.data_ptr(), wbytes, s)
    assert bwd() == 0                                      # all gradient pointers NULL: nothing to write, still fine
    assert bwd(wbytes=wb // 2) == -4
    assert bwd(dxyz=out.data_ptr()) == -2                  # d_xyz belongs to explicit points, d_z to rays
    assert bwd(grads=None) == -1
    v2 = N.pnr_views.from_buffer_copy(v); v2.latent[0] = None
    assert N.lib.pnr_point_mlp_train_fwd(C.byref(prm), C.byref(m), C.byref(v2), N.ptr(rays), N.ptr(z), K, None, None, P, P,
                                         N.ptr(out), tape.data_ptr(), tb, s) == -1        # needs the fp32 maps
    assert N.lib.pnr_composite_bwd(N.ptr(rays), N.ptr(z), out.data_ptr() + 4, n, K, 1, None, None, None, N.ptr(d_out), None, s) == -5
    assert N.lib.pnr_sample_fine_bwd(N.ptr(rays), None, n, 8, 4, 2, 0.01, None, 1, 0, N.ptr(z), N.ptr(z), N.ptr(z), s) == -1
    assert N.lib.pnr_point_mlp_train_fwd(C.byref(prm), C.byref(m), C.byref(v), N.ptr(rays), N.ptr(z), K, None, None, 0, 1,
                                         N.ptr(out), None, 0, s) in (0, -2)               # empty batch: no launch
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", ["full_ns1", "full_ns3", "full_multiscale_ns2", "tiny_ns2_codeview"])
def test_bf16_product_mode_gradients(name):
    """train_precision='bf16': GEMM products on the bf16 MFMA (fp32 accumulate, fp32 tape).  No reference numerics exist
    for it (the reference's AMP is fp16 autocast), so the bound is against the fp32 path on the same inputs.  Rounding
    28 chained GEMMs of a 5-block ReLU net to 8 mantissa bits moves each gradient tensor by 5-8 % in l2 on these
    256-point batches (a CPU emulation that rounds only the FORWARD operands already moves them by 8-10 % on the
    unmodified fixtures, whose sigma head is scaled x20 — removed here); a wrong operand layout would give cosine ~ 0."""

    def run(prec):
        fx, spec, net, rend = hu.setup(name)
        net.train()
        net.train_precision = prec
        with torch.no_grad():
            for m in (net.mlp_coarse, net.mlp_fine):
                if m is not None:
                    m.lin_out.weight[3] /= 20.0
                    m.lin_out.bias[3] = (m.lin_out.bias[3] - 4.0) / 20.0 + 1.0
        maps = [torch.from_numpy(x).cuda().requires_grad_(True) for x in gu.make_latents(spec)]
        net.encoder.set_latents(maps)
        out = rend(net, torch.from_numpy(fx["rays"]).cuda(), want_weights=True)
        G = {k: torch.from_numpy(v).cuda() for k, v in gu.make_loss_weights(spec).items()}
        # coarse pass only: the fine pass resamples from low-precision weights (discontinuous)
        loss = (out.coarse.rgb * G["coarse_rgb"]).sum() + (out.coarse.weights * G["coarse_weights"]).sum()
        loss.backward()
        grads = {"coarse." + k: p.grad.clone() for k, p in net.mlp_coarse.named_parameters()}
        grads.update({f"latent.{i}": m.grad.clone() for i, m in enumerate(maps)})
        return out.coarse.rgb.detach().clone(), grads

    rgb32, g32 = run("fp32")
    rgb16, g16 = run("bf16")
    assert float((rgb32 - rgb16).abs().max()) <= 1e-2
    assert float(((rgb32 - rgb16) ** 2).mean()) <= 1e-5           # >= 50 dB
    for k in g32:
        a, b = g16[k].double().flatten(), g32[k].double().flatten()
        nb = float(b.norm())
        if nb == 0:
            continue
        rel = float((a - b).norm()) / nb
        cos = float((a @ b) / (a.norm() * b.norm()))
        assert rel <= 0.15 and cos >= 0.99, (k, rel, cos)


def test_frozen_weights_still_give_bias_gradients():
    """needs_input_grad False for a weight, True for its bias: the bias gradient must still be the reference's
    (the GEMM kernels need dW; the bias-only request takes the column-sum kernel)."""
    name = "tiny_ns2_codeview"
    gfx = gu.load_grad_fixture(name)
    fx, spec, net, rend = hu.setup(name)
    net.train()
    for mlp in (net.mlp_coarse, net.mlp_fine):
        for k, p in mlp.named_parameters():
            if k.endswith("weight"):
                p.requires_grad_(False)
    rays = torch.from_numpy(fx["rays"]).cuda()
    out = rend(net, rays, want_weights=True)
    G = {k: torch.from_numpy(v).cuda() for k, v in gu.make_loss_weights(spec).items()}
    loss = sum((out[t].rgb * G[f"{t}_rgb"]).sum() + (out[t].depth * G[f"{t}_depth"]).sum()
               + (out[t].weights * G[f"{t}_weights"]).sum() for t in ("coarse", "fine"))
    loss.backward()
    grads = {}
    for which, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)):
        for k, p in mlp.named_parameters():
            if k.endswith("weight"):
                assert p.grad is None
            else:
                assert p.grad is not None and float(p.grad.abs().max()) > 0, k
                grads[f"{which}.{k}"] = p.grad.cpu().numpy()
    compare_grads(grads, {k: v for k, v in gfx.items() if k.split("__")[0] in grads or k == "loss"}, RTOL, name)


@pytest.mark.parametrize("train_precision", ["fp32", "bf16", "bf16x3"])
def test_weight_gradients_are_run_to_run_identical(train_precision):
    """Every row split of a dW GEMM writes its own partial slice and the slices are summed in a fixed order (no fp32
    atomics race): the same step twice gives BIT-identical MLP weight / bias gradients, at a realistic step size
    (2 x 128 rays x (64+32) samples, d_hidden 512, 2 views: tens of row splits per GEMM)."""
    spec = gu._case(seed=78, d_hidden=512, lat=[(256, 16, 16)], image=(128, 128), focal=131.25, NS=2, SB=2, N=128,
                    Kc=64, Kf=32, Kfd=16)
    rays_np, poses_np = gu.make_inputs(spec)
    net = hu.build_net(spec, poses_np).train()
    net.train_precision = train_precision
    rend = hu.build_renderer(spec)
    rend.forced_seed = 4321
    rays = torch.from_numpy(rays_np).cuda()
    G = torch.randn(2, 128, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))

    def run():
        net.zero_grad()
        out = rend(net, rays)
        (out.fine.rgb * G).sum().add((out.coarse.rgb * G).sum()).backward()
        return [p.grad.clone() for p in net.mlp_coarse.parameters()] + [p.grad.clone() for p in net.mlp_fine.parameters()]

    a, b = run(), run()
    assert all(float(x.abs().max()) > 0 for x in a)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("NS,SB,lat", [(1, 2, (256, 8, 8)), (3, 1, (256, 8, 8)), (2, 2, (64, 16, 16))])
def test_latent_map_gradients_are_run_to_run_identical(NS, SB, lat):
    """Latent maps that fit the LDS (the SRN / NMR shapes the reference trains on): per-block partial maps filled by
    channel-owning threads in point order + an ordered reduction — no float atomics, so the encoder's incoming gradient
    is BIT-identical from run to run, and it still matches the atomics-free reference arithmetic (checked against the
    gradient of a float64 oracle elsewhere in this file; here: two runs, and a third with the batch rendered twice as
    large proves the slices are really summed, not overwritten)."""
    spec = gu._case(seed=79, d_hidden=512, lat=[lat], image=(128, 128), focal=131.25, NS=NS, SB=SB, N=300,
                    Kc=32, Kf=16, Kfd=8)
    rays_np, poses_np = gu.make_inputs(spec)
    net = hu.build_net(spec, poses_np).train()
    rend = hu.build_renderer(spec)
    rend.forced_seed = 99
    rays = torch.from_numpy(rays_np).cuda()
    G = torch.randn(SB, 300, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    base = [torch.from_numpy(x).cuda() for x in gu.make_latents(spec)]

    def run():
        maps = [m.clone().requires_grad_(True) for m in base]
        net.encoder.set_latents(maps)
        out = rend(net, rays)
        (out.fine.rgb * G).sum().add((out.coarse.rgb * G).sum()).backward()
        return maps[0].grad.clone()

    a, b = run(), run()
    assert float(a.abs().max()) > 0 and torch.equal(a, b)
    # 300 rays x 48 samples = 14400 points per object = 57 blocks of 256 points per view: the reduction really spans slices
    assert a.shape == base[0].shape


@pytest.mark.parametrize("NS,SB,lat,cv,tiny_loss", [
    (3, 1, [(256, 19, 25)], False, False),                                            # BASELINE cfg 4's map (DTU)
    (2, 1, [(64, 32, 32), (64, 32, 32), (128, 16, 16), (256, 8, 8)], True, False),    # four levels (cfg 5 with the first pool)
    (2, 2, [(256, 19, 25)], False, True),                                             # two objects; a loss scaled by 2^-30
])
def test_large_and_multilevel_latent_map_gradients_are_run_to_run_identical(NS, SB, lat, cv, tiny_loss):
    """Maps beyond the LDS path (DTU's 19 x 25 x 256, every multi-scale shape): the tap contributions are summed with 64-bit
    FIXED-POINT integer atomics into workspace copies of the maps — integer addition commutes, so the result does not depend
    on arrival order — scaled by the gradient's own magnitude (2^-40 of its largest element), then added to d_latent.
    The encoder's incoming gradient is bit-identical from run to run for these shapes too (round 3: fp32 atomics, sum order
    = arrival order), still matches the reference's gradients (test_gradients_match_reference covers full_dtu_ns3 and
    full_multiscale_ns2 through this path), and a loss 2^-30 times smaller gives exactly the 2^-30-fold gradient."""
    spec = gu._case(seed=83, d_hidden=512, lat=lat, image=(400, 300) if len(lat) == 1 else (128, 128), focal=360.0 if len(lat) == 1 else 131.25,
                    NS=NS, SB=SB, N=200, Kc=32, Kf=16, Kfd=8, use_code_viewdirs=cv)
    rays_np, poses_np = gu.make_inputs(spec)
    net = hu.build_net(spec, poses_np).train()
    rend = hu.build_renderer(spec)
    rend.forced_seed = 99
    rays = torch.from_numpy(rays_np).cuda()
    G = torch.randn(SB, 200, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    base = [torch.from_numpy(x).cuda() for x in gu.make_latents(spec)]

    def run(scale=1.0):
        maps = [m.clone().requires_grad_(True) for m in base]
        net.encoder.set_latents(maps)
        out = rend(net, rays)
        ((out.fine.rgb * G).sum().add((out.coarse.rgb * G).sum()) * scale).backward()
        return [m.grad.clone() for m in maps]

    a, b = run(), run()
    for x, y in zip(a, b):
        assert float(x.abs().max()) > 0 and torch.equal(x, y)
    if not tiny_loss:
        # a non-finite cotangent has no fixed-point image: it must surface in the map (a NaN), not vanish in the conversion
        Gbad = G.clone()
        Gbad[0, 7, 1] = float("nan")
        maps = [m.clone().requires_grad_(True) for m in base]
        net.encoder.set_latents(maps)
        out = rend(net, rays)
        (out.fine.rgb * Gbad).sum().backward()
        assert all(bool(torch.isnan(m.grad).any()) for m in maps)
    if tiny_loss:
        # a loss 2^-30 times smaller: every fp32 step of the backward scales exactly, and so must the fixed-point sums — their
        # scale follows the gradient's magnitude (a fixed scale would have rounded these away)
        c = run(2.0 ** -30)
        for x, y in zip(a, c):
            assert torch.equal(y * 2.0 ** 30, x)


@pytest.mark.parametrize("name", ["full_ns1", "full_ns3", "full_multiscale_ns2", "tiny_ns2_codeview", "tiny_max_combine"])
def test_bf16_mode_16bit_tape_gives_the_fp32_tapes_gradients_bit_for_bit(name):
    """train_precision='bf16' keeps the block inputs and fc_0 outputs on the tape as bf16 — the values its GEMMs stage
    anyway (same v_cvt_pk_bf16_f32 rounding, relu commutes with it, the relu masks only look at the sign) — so every
    output and every gradient equals the fp32-tape run of the same mode bit for bit, at ~0.6x the tape bytes."""
    import ctypes as C
    from pixel_nerf_multiscale_amd import _native as N

    def run(tape):
        fx, spec, net, rend = hu.setup(name)
        net.train()
        net.train_precision, net.train_tape = "bf16", tape
        maps = [torch.from_numpy(x).cuda().requires_grad_(True) for x in gu.make_latents(spec)]
        net.encoder.set_latents(maps)
        out = rend(net, torch.from_numpy(fx["rays"]).cuda(), want_weights=True)
        G = {k: torch.from_numpy(v).cuda() for k, v in gu.make_loss_weights(spec).items()}
        loss = sum((out[t].rgb * G[f"{t}_rgb"]).sum() + (out[t].weights * G[f"{t}_weights"]).sum() for t in ("coarse", "fine"))
        loss.backward()
        grads = {f"{w}.{k}": p.grad.clone() for w, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)) if mlp is not None
                 for k, p in mlp.named_parameters()}
        grads.update({f"latent.{i}": m.grad.clone() for i, m in enumerate(maps)})
        prm = net.params_struct(None, "bf16")
        prm.train_tape_fp32 = 1 if tape == "fp32" else 0
        v, _ = net.views_struct("fp32")
        m, _ = net.mlp_struct(net.mlp_coarse, "fp32")
        nbytes = N.lib.pnr_train_tape_bytes_for(C.byref(prm), C.byref(m), C.byref(v), 65536 * v.n_objs)     # a training-sized batch
        return out.fine.rgb.detach().clone(), grads, nbytes

    rgb16, g16, b16 = run("auto")
    rgb32, g32, b32 = run("fp32")
    assert torch.equal(rgb16, rgb32)
    for k in g32:
        # (multi-level latent maps included since round 4: their taps are summed in fixed point, order-independent)
        assert torch.equal(g16[k], g32[k]), k
    if "full" in name:
        # d_hidden 512: the 16-bit tape applies — 0.6x with one view; the fp32 per-view stream of a multi-view net stays, and the
        # bf16 operand copies of round 4's LDS-DMA GEMMs (weights, latent columns) ride on the tape too
        assert b16 < (0.75 if name == "full_ns1" else 0.95) * b32
    else:
        assert b16 <= b32
