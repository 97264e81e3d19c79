"""
Training path: the renderer under autograd, as train/train.py:324-346,382-410 uses the
reference (calc_losses -> renderer(net, rays, want_weights=True) -> loss.backward()).

Three torch.autograd.Functions, each a thin shell over a forward/backward pair of libpnr_hip.so entry
points (csrc/train_f32.hip) — no PyTorch arithmetic on the per-point data:

  PointMLP      pnr_point_mlp_train_fwd / pnr_point_mlp_bwd    PixelNeRFNet.forward (models.py.backup2:155-282)
  Composite     pnr_composite / pnr_composite_bwd              NeRFRenderer.composite (nerf.py:178-182,223-249)
  SampleFine    pnr_sample_fine / pnr_sample_fine_bwd          sample_fine* + cat + sort (nerf.py:120-161,285-295)

Gradients reach the MLP weights, the encoder's latent maps (and through them the ResNet trunk, which stays
a PyTorch module) and — because nerf.py:287-289 does not detach the depth-guided samples — the coarse depth
through the fine pass's sample positions.  Arithmetic is fp32 (the reference trains in fp32);
`net.train_precision = "bf16"` runs the GEMM products on the bf16 MFMA with fp32 accumulation, a 16-bit tape (the block inputs
and fc_0 outputs kept as the bf16 values those GEMMs stage anyway: gradients bit-identical to an fp32 tape's, which
`net.train_tape = "fp32"` selects) and fp32
results (the counterpart of the reference's AMP switch, train/train.py:385-398).
"""
import ctypes as C

import torch

from .. import _native as N
from ..model.models import mlp_tensors, views_from


def _mlp_struct_from(hdr, tensors):
    """pnr_mlp from the ordered parameter list of models.mlp_tensors()."""
    m = N.pnr_mlp()
    m.d_in, m.d_latent, m.d_hidden, m.d_out = hdr["d_in"], hdr["d_latent"], hdr["d_hidden"], hdr["d_out"]
    m.n_blocks, m.combine_layer, m.combine_type = hdr["n_blocks"], hdr["combine_layer"], N.COMBINE[hdr["combine_type"]]
    keep = [N.f32c(t.detach()) for t in tensors]
    it = iter(keep)
    m.lin_in_w, m.lin_in_b = N.ptr(next(it)), N.ptr(next(it))
    m.lin_out_w, m.lin_out_b = N.ptr(next(it)), N.ptr(next(it))
    for b in range(hdr["n_blocks"]):
        m.fc0_w[b], m.fc0_b[b] = N.ptr(next(it)), N.ptr(next(it))
        m.fc1_w[b], m.fc1_b[b] = N.ptr(next(it)), N.ptr(next(it))
    for b in range(hdr["n_lin_z"]):
        m.lin_z_w[b], m.lin_z_b[b] = N.ptr(next(it)), N.ptr(next(it))
    return m, keep


def _grads_struct_from(hdr, grads):
    g = N.pnr_mlp_grads()
    it = iter(grads)
    P = lambda t: None if t is None else t.data_ptr()
    g.lin_in_w, g.lin_in_b = P(next(it)), P(next(it))
    g.lin_out_w, g.lin_out_b = P(next(it)), P(next(it))
    for b in range(hdr["n_blocks"]):
        g.fc0_w[b], g.fc0_b[b] = P(next(it)), P(next(it))
        g.fc1_w[b], g.fc1_b[b] = P(next(it)), P(next(it))
    for b in range(hdr["n_lin_z"]):
        g.lin_z_w[b], g.lin_z_b[b] = P(next(it)), P(next(it))
    return g


def _train_params(net):
    prm = net.params_struct(None, net.train_precision)
    prm.train_tape_fp32 = 1 if getattr(net, "train_tape", "auto") == "fp32" else 0
    return prm


class PointMLP(torch.autograd.Function):
    """out (n_points, 4) = PixelNeRFNet.forward at points named by rays+z (a, b = rays (N,8), z (N,K)) or
    explicitly (a, b = xyz (SB*P,3), viewdirs (SB*P,3)).  Differentiable in b=z / a=xyz, the MLP parameters
    and the latent maps."""

    @staticmethod
    def forward(ctx, hdr, a, b, poses, focal, c, *tensors):
        n_par = hdr["n_params"]
        params, maps = tensors[:n_par], tensors[n_par:]
        dev = a.device
        a, b = N.f32c(a.detach()), N.f32c(b.detach())
        m, k1 = _mlp_struct_from(hdr, params)
        v, k2 = views_from(poses, focal, c, hdr["n_views"], maps, hdr["uv_scale"])
        rays_mode = hdr["rays_mode"]
        if rays_mode:
            n_rays, K = b.shape
            n_points = n_rays * K
        else:
            n_points, K = a.shape[0], 0
        if n_points % v.n_objs:
            raise ValueError("points do not divide evenly over the encoded objects")
        prm = hdr["prm"]
        tape = torch.empty(int(N.lib.pnr_train_tape_bytes_for(C.byref(prm), C.byref(m), C.byref(v), n_points)), dtype=torch.uint8,
                           device=dev)
        out = torch.empty(n_points, 4, device=dev)
        args = (N.ptr(a), N.ptr(b), K, None, None) if rays_mode else (None, None, 0, N.ptr(a), N.ptr(b))
        N.check(N.lib.pnr_point_mlp_train_fwd(C.byref(prm), C.byref(m), C.byref(v), *args, n_points,
                                              n_points // v.n_objs, N.ptr(out), tape.data_ptr(), tape.numel(),
                                              N.current_stream(dev)), "pnr_point_mlp_train_fwd")
        ctx.hdr = hdr
        ctx.save_for_backward(a, b, poses, focal, c, out, tape, *params, *maps)
        return out

    @staticmethod
    def backward(ctx, d_out):
        hdr = ctx.hdr
        a, b, poses, focal, c, out, tape = ctx.saved_tensors[:7]
        rest = ctx.saved_tensors[7:]
        n_par = hdr["n_params"]
        params, maps = rest[:n_par], rest[n_par:]
        dev = a.device
        m, k1 = _mlp_struct_from(hdr, params)
        v, k2 = views_from(poses, focal, c, hdr["n_views"], maps, hdr["uv_scale"])
        need = ctx.needs_input_grad            # (hdr, a, b, poses, focal, c, *params, *maps)
        need_par, need_map = need[6:6 + n_par], need[6 + n_par:]
        # one zero fill for all parameter gradients (the kernels accumulate with atomics)
        flat = torch.zeros(sum(p.numel() for p, nd in zip(params, need_par) if nd), device=dev)
        g_par, off = [], 0
        for p, nd in zip(params, need_par):
            g_par.append(flat[off:off + p.numel()].view(p.shape) if nd else None)
            off += p.numel() if nd else 0
        g_map = [torch.zeros(mp.shape, device=dev) if nd else None for mp, nd in zip(maps, need_map)]
        rays_mode = hdr["rays_mode"]
        n_points = out.shape[0]
        d_pos = None
        if rays_mode and need[2]:
            d_pos = torch.empty_like(b)
        elif not rays_mode and need[1]:
            d_pos = torch.empty_like(a)
        g = _grads_struct_from(hdr, g_par)
        dl = (C.c_void_p * N.PNR_MAX_LEVELS)(*[None if t is None else t.data_ptr() for t in g_map])
        ws = torch.empty(int(N.lib.pnr_train_bwd_workspace_bytes(C.byref(m), C.byref(v), n_points)), dtype=torch.uint8,
                         device=dev)
        d_out = N.f32c(d_out)
        K = b.shape[1] if rays_mode else 0
        args = (N.ptr(a), N.ptr(b), K, None, None) if rays_mode else (None, None, 0, N.ptr(a), N.ptr(b))
        N.check(N.lib.pnr_point_mlp_bwd(C.byref(hdr["prm"]), C.byref(m), C.byref(v), *args, n_points,
                                        n_points // v.n_objs, N.ptr(out), N.ptr(d_out), tape.data_ptr(), tape.numel(),
                                        C.byref(g), dl,
                                        None if (rays_mode or d_pos is None) else N.ptr(d_pos),
                                        N.ptr(d_pos) if (rays_mode and d_pos is not None) else None,
                                        ws.data_ptr(), ws.numel(), N.current_stream(dev)), "pnr_point_mlp_bwd")
        g_a = d_pos if not rays_mode else None
        g_b = d_pos if rays_mode else None
        return (None, g_a, g_b, None, None, None, *g_par, *g_map)


class Composite(torch.autograd.Function):
    """(weights, rgb, depth) = composite(rays, z, rgbsigma); differentiable in rgbsigma and z."""

    @staticmethod
    def forward(ctx, rays, z, rgbs, white_bkgd):
        rays, z, rgbs = N.f32c(rays.detach()), N.f32c(z.detach()), N.f32c(rgbs.detach())
        B, K = z.shape
        dev = rays.device
        weights, rgb, depth = torch.empty(B, K, device=dev), torch.empty(B, 3, device=dev), torch.empty(B, device=dev)
        N.check(N.lib.pnr_composite(N.ptr(rays), N.ptr(z), N.ptr(rgbs), B, K, int(bool(white_bkgd)), N.ptr(weights),
                                    N.ptr(rgb), N.ptr(depth), N.current_stream(dev)), "pnr_composite")
        ctx.white = int(bool(white_bkgd))
        ctx.save_for_backward(rays, z, rgbs)
        return weights, rgb, depth

    @staticmethod
    def backward(ctx, d_w, d_rgb, d_depth):
        rays, z, rgbs = ctx.saved_tensors
        B, K = z.shape
        dev = rays.device
        P = lambda t: None if t is None else N.ptr(N.f32c(t))
        keep = [None if t is None else N.f32c(t) for t in (d_w, d_rgb, d_depth)]
        d_rgbs = torch.empty_like(rgbs)
        d_z = torch.empty_like(z) if ctx.needs_input_grad[1] else None
        N.check(N.lib.pnr_composite_bwd(N.ptr(rays), N.ptr(z), N.ptr(rgbs), B, K, ctx.white,
                                        *[None if t is None else N.ptr(t) for t in keep],
                                        N.ptr(d_rgbs), None if d_z is None else N.ptr(d_z), N.current_stream(dev)),
                "pnr_composite_bwd")
        return None, d_z, d_rgbs, None


class SampleFine(torch.autograd.Function):
    """z_sorted (N, Kc+Kf) = sort(cat(z_coarse, importance samples, depth samples)); differentiable in depth."""

    @staticmethod
    def forward(ctx, rend, rays, z_coarse, weights, depth, seed, noise):
        dev = rays.device
        rays, zc = N.f32c(rays.detach()), N.f32c(z_coarse.detach())
        w, d = N.f32c(weights.detach()), N.f32c(depth.detach())
        B = rays.shape[0]
        z = torch.empty(B, rend.n_coarse + rend.n_fine, device=dev)
        u, r, g = (None if noise is None or noise.get(k) is None else N.f32c(noise[k], dev) for k in ("u", "r", "g"))
        P = lambda t: None if t is None else N.ptr(t)
        cfg = (int(rend.n_coarse), int(rend.n_fine), int(rend.n_fine_depth), float(rend.depth_std), int(bool(rend.lindisp)),
               int(seed), int(rend.ray_index_base))
        N.check(N.lib.pnr_sample_fine(N.ptr(rays), N.ptr(zc), N.ptr(w), N.ptr(d), B, cfg[0], cfg[1], cfg[2], cfg[3], cfg[4],
                                      P(u), P(r), P(g), cfg[5], cfg[6], N.ptr(z), N.current_stream(dev)), "pnr_sample_fine")
        ctx.cfg = cfg
        ctx.g = g
        ctx.save_for_backward(rays, d, z)
        return z

    @staticmethod
    def backward(ctx, d_z):
        rays, d, z = ctx.saved_tensors
        Kc, Kf, Kfd, std, _, seed, base = ctx.cfg
        if Kfd == 0 or not ctx.needs_input_grad[4]:
            return (None,) * 7
        d_depth = torch.empty_like(d)
        d_z = N.f32c(d_z)
        N.check(N.lib.pnr_sample_fine_bwd(N.ptr(rays), N.ptr(d), rays.shape[0], Kc, Kf, Kfd, std,
                                          None if ctx.g is None else N.ptr(ctx.g), seed, base, N.ptr(z), N.ptr(d_z),
                                          N.ptr(d_depth), N.current_stream(rays.device)), "pnr_sample_fine_bwd")
        return None, None, None, None, d_depth, None, None


# ---------------------------------------------------------------------------------------------------------------
def _header(net, mlp, rays_mode):
    return dict(d_in=mlp.d_in, d_latent=mlp.d_latent, d_hidden=mlp.d_hidden, d_out=mlp.d_out, n_blocks=mlp.n_blocks,
                combine_layer=mlp.combine_layer, combine_type=mlp.combine_type,
                n_lin_z=len(mlp.lin_z) if mlp.d_latent else 0, n_params=len(mlp_tensors(mlp)),
                n_views=int(net.num_views_per_obj), rays_mode=rays_mode, prm=_train_params(net),
                uv_scale=net.uv_scales())


def point_mlp_rays(net, mlp, rays, z):
    """(N, K, 4) network outputs at the samples z (N,K) of rays (N,8), under autograd."""
    hdr = _header(net, mlp, True)
    out = PointMLP.apply(hdr, rays, z, net.poses, net.focal, net.c, *mlp_tensors(mlp), *net.latent_maps_for_grad())
    return out.reshape(z.shape[0], z.shape[1], 4)


def point_mlp_points(net, mlp, xyz, viewdirs):
    """(SB, P, 4) network outputs at explicit points, under autograd (PixelNeRFNet.forward in training)."""
    SB, P, _ = xyz.shape
    hdr = _header(net, mlp, False)
    out = PointMLP.apply(hdr, xyz.reshape(-1, 3), viewdirs.reshape(-1, 3), net.poses, net.focal, net.c,
                         *mlp_tensors(mlp), *net.latent_maps_for_grad())
    return out.reshape(SB, P, 4)


def render_train(rend, net, rays, want_weights):
    """NeRFRenderer.forward (nerf.py:251-303) with every stage differentiable where the reference's is."""
    from ..util import AttrDict
    SB, B, _ = rays.shape
    r = N.f32c(rays).reshape(-1, 8)
    if net.poses.shape[0] != SB * int(net.num_views_per_obj):
        raise ValueError(f"rays has {SB} objects but encode() saw {net.poses.shape[0] // int(net.num_views_per_obj)}")
    seed = rend._seed()
    noise = rend.fixed_noise

    def sigma_noise(out):        # nerf.py:225-226, training only
        if rend.training and rend.noise_std > 0.0:
            return torch.cat([out[..., :3], out[..., 3:] + torch.randn_like(out[..., 3:]) * rend.noise_std], dim=-1)
        return out

    z_c = rend.sample_coarse(r, seed)
    out_c = sigma_noise(point_mlp_rays(net, net.mlp_coarse, r, z_c))
    comp_c = Composite.apply(r, z_c, out_c, rend.white_bkgd)
    res = AttrDict(coarse=rend._format_outputs(comp_c, SB, want_weights))
    if rend.using_fine:
        z_f = SampleFine.apply(rend, r, z_c, comp_c[0], comp_c[2], seed, noise)
        mlp_f = net.mlp_fine if net.mlp_fine is not None else net.mlp_coarse
        out_f = sigma_noise(point_mlp_rays(net, mlp_f, r, z_f))
        res.fine = rend._format_outputs(Composite.apply(r, z_f, out_f, rend.white_bkgd), SB, want_weights)
    return res
