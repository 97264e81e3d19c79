"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise ShardedRenderer's ray-range sharding, the
per-frame seed derivation and the one all_gather per frame.  The per-rank renderer is a deterministic stand-in that,
like the HIP kernels, depends only on (ray, GLOBAL ray index, seed) — so the gathered frame must equal the
unsharded one bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pixel_nerf_multiscale_amd.parallel import ShardedRenderer, frame_seed, shard_range


def fake_render(rays, base, seed):
    n = rays.shape[1]
    gidx = torch.arange(base, base + n, dtype=torch.float64)
    s = float(seed % 1000003) / 1000003.0
    rgb = torch.stack([torch.sin(gidx * 0.37 + s), rays[0, :, 3].double() * 0.5 + s, torch.cos(gidx * 0.11)], -1).float()
    depth = (rays[0, :, 6].double() + gidx * 1e-3 + s).float()
    return rgb[None], depth[None]


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        rays = torch.rand(1, B, 8, generator=g)
        sr = ShardedRenderer(fake_render)                  # base seed broadcast from rank 0
        outs = [sr(rays) for _ in range(2)]                # two frames: different per-frame seeds
        q.put((rank, sr.base_seed, [(r.numpy().copy(), d.numpy().copy()) for r, d in outs]))   # by value, not shared storage
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# world 8 = the node the scaling bench runs on: BASELINE cfg 4's 120 000-ray DTU frame (15 000 rays per rank), and fewer rays
# than ranks (three ranks render nothing and still take part in the collective)
@pytest.mark.parametrize("world,B", [(2, 64), (2, 37), (3, 10), (2, 1), (8, 120000), (8, 5)])
def test_sharded_equals_unsharded(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seeds = {r[1] for r in res}
    assert len(seeds) == 1                                  # every rank uses rank 0's base seed
    base = seeds.pop()
    g = torch.Generator().manual_seed(7)
    rays = torch.rand(1, B, 8, generator=g)
    for frame in range(2):
        ref_rgb, ref_depth = fake_render(rays, 0, frame_seed(base, frame))
        for rank, _, outs in res:
            rgb, depth = (torch.from_numpy(t) for t in outs[frame])
            assert rgb.shape == (1, B, 3) and depth.shape == (1, B)
            assert torch.equal(rgb, ref_rgb) and torch.equal(depth, ref_depth), (rank, frame)


def test_shard_range_covers_all_rays():
    for n in (0, 1, 5, 64, 120000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r)[:2] for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
    assert frame_seed(5, 0) != frame_seed(5, 1) and frame_seed(5, 0) == frame_seed(5, 0)


def test_single_process_passthrough():
    rays = torch.rand(1, 9, 8)
    sr = ShardedRenderer(fake_render, base_seed=3)
    rgb, depth = sr(rays)
    ref = fake_render(rays, 0, frame_seed(3, 0))
    assert torch.equal(rgb, ref[0]) and torch.equal(depth, ref[1])


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pixel_nerf_multiscale_amd.parallel import allreduce_gradients
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
                  torch.nn.Parameter(torch.zeros(2, 2), requires_grad=False), torch.nn.Parameter(torch.zeros(4))]
        params[0].grad = torch.full((5, 3), float(rank + 1))
        params[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
        if rank == 0:
            params[3].grad = torch.ones(4)                 # missing on the other ranks: counts as zeros
        n = allreduce_gradients(params, bucket_bytes=64)   # tiny buckets: several collectives
        q.put((rank, n, [None if p.grad is None else p.grad.numpy().copy() for p in params]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allreduce_gradients_averages_over_ranks(world):
    """Data-parallel training step: per-rank gradients from the HIP backward are averaged in flat buckets."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean_scale = sum(range(1, world + 1)) / world
    for rank, n, grads in res:
        assert n >= 2
        assert (grads[0] == mean_scale).all()
        assert torch.allclose(torch.from_numpy(grads[1]), torch.arange(7, dtype=torch.float32) * mean_scale)
        assert grads[2] is None
        assert torch.allclose(torch.from_numpy(grads[3]), torch.full((4,), 1.0 / world))


# ----------------------------------------------------------------------------- bind_parallel as the drop-in (a16)
def _cpu_renderer_cls():
    """NeRFRenderer's surface with a deterministic CPU body that, like the HIP kernels, depends only on
    (ray, GLOBAL ray index, seed) — so the real bind_parallel -> _ShardedRenderWrapper -> ShardedRenderer path runs here."""
    from pixel_nerf_multiscale_amd import NeRFRenderer
    from pixel_nerf_multiscale_amd.util import AttrDict

    class CpuRenderer(NeRFRenderer):
        def forward(self, model, rays, want_weights=False):
            rgb, depth = fake_render(rays, self.ray_index_base, self.forced_seed)
            out = AttrDict(coarse=AttrDict(rgb=rgb * 0.5, depth=depth * 0.5))
            if self.using_fine:
                out.fine = AttrDict(rgb=rgb, depth=depth)
            if want_weights:          # (1, n, K) per level, a function of the global ray index like everything else
                idx = torch.arange(self.ray_index_base, self.ray_index_base + rays.shape[1], dtype=torch.float32)
                out.coarse.weights = (idx[None, :, None] + torch.arange(self.n_coarse)[None, None, :] * 0.25)
                if self.using_fine:
                    out.fine.weights = (idx[None, :, None] * 2 + torch.arange(self.n_coarse + self.n_fine)[None, None, :] * 0.5)
            return out
    return CpuRenderer


def _bind_worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        all_rays = torch.rand(B, 8, generator=g)
        rend = _cpu_renderer_cls()(n_coarse=4, n_fine=2)
        # the reference call site: eval/eval.py:151  render_par = renderer.bind_parallel(net, args.gpu_id, simple_output=True)
        render_par = rend.bind_parallel(None, list(range(world)), simple_output=True)
        rgb, depth = render_par(all_rays[None])                        # eval/eval.py:280
        full = rend.bind_parallel(None, list(range(world)), simple_output=False)
        d = full(all_rays[None])
        dw = full(all_rays[None], want_weights=True)                   # nested output with per-sample weights (nerf.py:33-41)
        assert set(dw["fine"]) == {"rgb", "depth", "weights"} and tuple(dw["fine"]["weights"].shape) == (1, B, 6)
        assert torch.equal(dw["coarse"]["weights"][0, :, 0], torch.arange(B, dtype=torch.float32))
        assert torch.equal(dw["fine"]["weights"][0, :, 5], torch.arange(B, dtype=torch.float32) * 2 + 2.5)
        q.put((rank, render_par.sharded.base_seed, rgb.numpy().copy(), depth.numpy().copy(),
               full.sharded.base_seed, {k: {kk: vv.numpy().copy() for kk, vv in v.items()} for k, v in d.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 50), (3, 7)])
def test_bind_parallel_shards_under_a_process_group(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bind_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(11)
    rays = torch.rand(1, B, 8, generator=g)
    for rank, seed1, rgb, depth, seed2, d in res:
        ref_rgb, ref_depth = fake_render(rays, 0, frame_seed(seed1, 0))
        assert rgb.shape == (1, B, 3) and depth.shape == (1, B)
        assert torch.equal(torch.from_numpy(rgb), ref_rgb) and torch.equal(torch.from_numpy(depth), ref_depth)   # fine level
        r2, d2 = fake_render(rays, 0, frame_seed(seed2, 0))
        assert set(d) == {"coarse", "fine"}
        assert torch.equal(torch.from_numpy(d["fine"]["rgb"]), r2) and torch.equal(torch.from_numpy(d["coarse"]["depth"]), d2 * 0.5)


def test_bind_parallel_without_process_group_warns_and_renders_on_one_device():
    rend = _cpu_renderer_cls()(n_coarse=4, n_fine=0)
    with pytest.warns(UserWarning, match="torch.distributed.run"):
        render_par = rend.bind_parallel(None, [0, 1, 2, 3], simple_output=True)
    rend.forced_seed = 5
    rays = torch.rand(1, 6, 8)
    rgb, depth = render_par(rays)
    ref = fake_render(rays, 0, 5)
    assert torch.equal(rgb, ref[0] * 0.5) and torch.equal(depth, ref[1] * 0.5)
    rgb0, depth0 = render_par(torch.zeros(0, 8))            # zero-ray early-out (nerf.py:23-27)
    assert rgb0.shape == (0, 3) and depth0.shape == (0,)


# ----------------------------------------------------------------------------- training under the same binding (train/train.py:171,331)
class _TinyNet(torch.nn.Module):
    """Stand-in with PixelNeRFNet's training switch (wants_grad) and a differentiable CPU body."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([[0.3, -0.2, 0.5], [0.1, 0.4, -0.6], [0.7, 0.2, 0.1]]))

    def wants_grad(self, *inputs):
        return self.training and torch.is_grad_enabled()


def _diff_renderer_cls():
    from pixel_nerf_multiscale_amd import NeRFRenderer
    from pixel_nerf_multiscale_amd.util import AttrDict

    class DiffRenderer(NeRFRenderer):
        def forward(self, model, rays, want_weights=False):
            rgb = torch.tanh(rays[..., 3:6] @ model.w)
            return AttrDict(coarse=AttrDict(rgb=rgb, depth=rays[..., 6] * model.w.sum()))
    return DiffRenderer


def _train_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pixel_nerf_multiscale_amd.parallel import allreduce_gradients
        g = torch.Generator().manual_seed(3)
        batch = torch.rand(1, 8 * world, 8, generator=g)               # the concatenated batch; rank r trains on its slice
        net = _TinyNet()
        rend = _diff_renderer_cls()(n_coarse=4, n_fine=0)
        render_par = rend.bind_parallel(net, list(range(world))).eval()  # train.py:171; .eval() reaches the net as well ...
        net.train()                                                      # ... and the trainer switches it back per step
        mine = batch[:, 8 * rank:8 * (rank + 1)]
        out = render_par(mine, want_weights=True)                        # train.py:331: rank-local, differentiable
        assert out["coarse"]["rgb"].requires_grad and tuple(out["coarse"]["rgb"].shape) == (1, 8, 3)
        loss = out["coarse"]["rgb"].square().mean() + out["coarse"]["depth"].mean()
        loss.backward()
        allreduce_gradients(list(net.parameters()))                      # the step hook
        # eval-mode calls through the SAME binding are sharded again
        net.eval()
        with torch.no_grad():
            full = render_par(batch)
        q.put((rank, net.w.grad.numpy().copy(), full["coarse"]["rgb"].numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_training_call_is_rank_local_and_averaged_gradients_equal_the_whole_batch(world):
    """Two (three) ranks, each back-propagating its own slice through the binding bind_parallel returns, then
    allreduce_gradients: the result equals the single-process gradient of the concatenated batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(3)
    batch = torch.rand(1, 8 * world, 8, generator=g)
    net = _TinyNet()
    rend = _diff_renderer_cls()(n_coarse=4, n_fine=0)
    out = rend(net, batch)
    (out.coarse.rgb.square().mean() + out.coarse.depth.mean()).backward()
    for rank, grad, full in res:
        assert torch.allclose(torch.from_numpy(grad), net.w.grad, rtol=1e-5, atol=1e-7), rank
        assert torch.allclose(torch.from_numpy(full), out.coarse.rgb.detach(), rtol=0, atol=0), rank
