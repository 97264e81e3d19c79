"""Maximum sizes (the edge case the reference never meets — its chunking caps a call at eval_batch_size points,
render/nerf.py:195-201; this library takes a frame in one launch): ONE pnr_point_mlp call over 2^31 + 512 points of two
objects.  Point indices beyond 2^31, the kernel's 64-bit division paths (object of a point, ray of a point: div_pts in
point_mfma.hip, not taken below 2^31 - 1 points) and the 16-B-per-point output at 34 GB offsets are exercised nowhere
else.  Size-independent check: the network is a per-point function, so any slice of the big call must be BIT-identical to
a small call on the same rays — taken at the start of object 0 and at the end of object 1 (global points past 2^31)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["bf16"])
def test_one_call_over_two_to_the_31_points(prec):
    import golden_util as gu
    from hip_util import build_net
    from pixel_nerf_multiscale_amd import _native as N
    free, total = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip("needs ~100 GB of free device memory")
    K, SB = 128, 2
    B = 2 ** 23 + 2                                  # rays per object: B*K = 2^30 + 256 points each, 2^31 + 512 in all
    n_small = 300
    spec = dict(gu.CASES["full_ns1"])
    spec.update(SB=SB, NS=1, N=4, seed=77)
    _, poses = gu.make_inputs(spec)
    net = build_net(spec, poses, "cuda", prec)
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    n = SB * B
    rays = torch.empty(n, 8, device=dev)
    rays[:, :3] = torch.randn(n, 3, device=dev, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 1.3], device=dev)
    d = torch.randn(n, 3, device=dev, generator=g)
    rays[:, 3:6] = d / d.norm(dim=1, keepdim=True)
    del d
    rays[:, 6], rays[:, 7] = 0.8, 1.8
    z = torch.rand(n, K, device=dev, generator=g) + 0.8
    prm = net.params_struct(None, prec)
    v, keep_v = net.views_struct(prec)
    m, keep_m = net.mlp_struct(net.mlp_coarse, prec, v)
    assert v.n_objs == SB
    n_points = n * K
    assert n_points >= 2 ** 31 - 1                   # the condition of the kernel's 64-bit paths

    def call(r, zz, b):
        pts = r.shape[0] * K
        out = torch.empty(pts, 4, device=dev)
        ws = torch.empty(N.lib.pnr_workspace_bytes(C.byref(prm), C.byref(m), C.byref(v), 0), dtype=torch.uint8, device=dev)
        N.check(N.lib.pnr_point_mlp(C.byref(prm), C.byref(m), C.byref(v), N.ptr(r), N.ptr(zz), K, None, None, pts, b * K,
                                    N.ptr(out), ws.data_ptr(), ws.numel(), N.current_stream(dev)), "pnr_point_mlp")
        return out

    big = call(rays, z, B)
    torch.cuda.synchronize()
    # the same rays in a small call: object 0's first rays and object 1's last ones
    sel = torch.cat([torch.arange(0, n_small, device=dev), torch.arange(n - n_small, n, device=dev)])
    small = call(rays[sel].contiguous(), z[sel].contiguous(), n_small)
    big3 = big.view(n, K, 4)
    assert torch.isfinite(small).all()
    assert torch.equal(big3[:n_small].reshape(-1, 4), small[: n_small * K]), "object 0, first rays"
    assert torch.equal(big3[n - n_small:].reshape(-1, 4), small[n_small * K:]), "object 1, last rays (points past 2^31)"
    # a slice that straddles the object boundary (the object index of a point changes inside a tile's range)
    mid = torch.cat([torch.arange(B - n_small, B, device=dev), torch.arange(B, B + n_small, device=dev)])
    small2 = call(rays[mid].contiguous(), z[mid].contiguous(), n_small)
    assert torch.equal(big3[B - n_small:B + n_small].reshape(-1, 4), small2), "around the object boundary"
    # nothing unwritten or non-finite anywhere (sampled: a full isfinite pass would allocate another 8 GB)
    step = 4099
    assert torch.isfinite(big3[::step]).all() and float(big3[::step, :, :3].min()) >= 0.0 and float(big3[::step, :, :3].max()) <= 1.0
    del big, big3, rays, z
    torch.cuda.empty_cache()


def test_one_render_call_over_two_to_the_31_points():
    """The fused render launch (rays in, pixels out) on a (2^24 + 4)-ray x 128-sample batch: the rgb-sigma / z workspaces
    are indexed past 2^31 points (34 GB), every workgroup composites its own rays.  With the in-kernel noise keyed by the
    global ray index, the last rays of the big call equal a small call on those rays with ray_index_base set."""
    import golden_util as gu
    from hip_util import build_net, build_renderer
    free, total = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip("needs ~100 GB of free device memory")
    n, n_small = 2 ** 24 + 4, 300
    spec = dict(gu.CASES["full_ns1"])
    spec.update(SB=1, NS=1, N=4, seed=78, Kc=128, Kf=0, Kfd=0)
    _, poses = gu.make_inputs(spec)
    net = build_net(spec, poses, "cuda", "bf16")
    rend = build_renderer(spec)
    rend.fixed_noise = None
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(4)
    rays = torch.empty(1, n, 8, device=dev)
    rays[0, :, :3] = torch.randn(n, 3, device=dev, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 1.3], device=dev)
    d = torch.randn(n, 3, device=dev, generator=g)
    rays[0, :, 3:6] = d / d.norm(dim=1, keepdim=True)
    del d
    rays[0, :, 6], rays[0, :, 7] = 0.8, 1.8
    rend.forced_seed = 99
    big = rend(net, rays, want_weights=True)
    torch.cuda.synchronize()
    assert tuple(big.coarse.rgb.shape) == (1, n, 3)
    for lo in (0, n // 2 + 1, n - n_small):
        rend.ray_index_base = lo
        small = rend(net, rays[:, lo:lo + n_small].contiguous(), want_weights=True)
        for k in ("rgb", "depth", "weights"):
            assert torch.equal(big.coarse[k][:, lo:lo + n_small], small.coarse[k]), (lo, k)
    rend.ray_index_base = 0
    assert torch.isfinite(big.coarse.rgb).all() and torch.isfinite(big.coarse.depth).all()
    del big, rays, net
    torch.cuda.empty_cache()
