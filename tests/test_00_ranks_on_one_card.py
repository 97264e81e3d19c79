"""The multi-rank render path on the real HIP renderer (SURVEY §8 e, a16): two ranks, one process each, BOTH on cuda:0
(the GPU box has one card and RCCL refuses duplicate devices, so the process group is gloo and the gathered message
travels through the host — everything else is the production path: NeRFRenderer.bind_parallel under an initialised
process group, parallel.ShardedRenderer, per-rank ray ranges, noise keyed by the global ray index).  The frame every rank
returns must equal the unsharded render bit for bit.  Runs first (file name) because the workers have to be started
before this process initialises the GPU."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, prec, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden_util as gu
        from hip_util import setup
        from pixel_nerf_multiscale_amd.parallel import frame_seed
        fx, spec, net, rend = setup(name, precision=prec)
        rend.fixed_noise = None
        W, H = spec["image"]
        g = torch.Generator().manual_seed(11)
        tgt = gu.pose_spherical(65.0, -25.0, spec["radius"])
        pix = torch.randperm(W * H, generator=g)[:1001].numpy()                      # odd count: ragged last shard
        rays = torch.from_numpy(gu.pinhole_rays(tgt, W, H, spec["focal"], spec["z_near"], spec["z_far"], pix))[None].cuda()
        render_par = rend.bind_parallel(net, [0, 1], simple_output=True).eval()      # the call site of eval/eval.py:151
        assert type(render_par).__name__ == "_ShardedRenderWrapper"
        rgb, depth = render_par(rays)                                                # eval/eval.py:280
        rend.forced_seed = frame_seed(render_par.sharded.base_seed, 0)               # the same frame on one rank
        ref = rend(net, rays)
        lvl = ref.fine if rend.using_fine else ref.coarse
        ok = bool(torch.equal(rgb, lvl.rgb)) and bool(torch.equal(depth, lvl.depth))
        # nested output with per-sample weights (nerf.py:33-41): they travel in the same all_gather
        full = rend.bind_parallel(net, [0, 1], simple_output=False).eval()
        dw = full(rays, want_weights=True)
        rend.forced_seed = frame_seed(full.sharded.base_seed, 0)
        refw = rend(net, rays, want_weights=True)
        for lv in dw:
            for k in ("rgb", "depth", "weights"):
                ok = ok and bool(torch.equal(dw[lv][k], refw[lv][k]))
        q.put((rank, ok, float(rgb.abs().sum())))
    finally:
        dist.destroy_process_group()


def _worker_sb2(rank, world, port, prec, q):
    """Several objects per call (nn.DataParallel(dim=1) cuts every object's rays: reference render/nerf.py:367-371) + the
    training call of train/train.py:171,331 under the same binding."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden_util as gu
        from hip_util import build_net, build_renderer
        from pixel_nerf_multiscale_amd.parallel import allreduce_gradients, frame_seed
        spec = dict(gu.CASES["full_ns1"]); spec.update(SB=2, NS=2, N=333, seed=91)     # odd count: ragged last shard
        rays_np, poses = gu.make_inputs(spec)
        rays = torch.from_numpy(rays_np).cuda()                                       # (2, 333, 8)
        net = build_net(spec, poses, "cuda", prec)
        rend = build_renderer(spec)
        ok = True
        for simple in (True, False):
            par = rend.bind_parallel(net, [0, 1], simple_output=simple).eval()
            out = par(rays, want_weights=not simple)
            rend.forced_seed = frame_seed(par.sharded.base_seed, 0)
            ref = rend(net, rays, want_weights=not simple)
            rend.forced_seed = None
            if simple:
                ok = ok and bool(torch.equal(out[0], ref.fine.rgb)) and bool(torch.equal(out[1], ref.fine.depth))
                ok = ok and tuple(out[0].shape) == (2, 333, 3)
            else:
                for lv in ("coarse", "fine"):
                    for k in ("rgb", "depth", "weights"):
                        ok = ok and bool(torch.equal(out[lv][k], ref[lv][k]))
        # training under the same binding: rank-local differentiable render (no sharding), gradients averaged by the step hook
        net32 = build_net(spec, poses, "cuda", "fp32")
        par = rend.bind_parallel(net32, [0, 1]).eval()           # train.py:171 (.eval() on the wrapper; the trainer sets net.train())
        net32.train()
        torch.manual_seed(5)
        sub = rays[:, rank * 64:(rank + 1) * 64].contiguous()    # every rank its own batch
        rend.forced_seed = 1234 + rank
        d = par(sub, want_weights=True)
        loss = d["fine"]["rgb"].square().mean() + d["coarse"]["rgb"].square().mean()
        loss.backward()
        rend.forced_seed = None
        g_local = net32.mlp_coarse.lin_out.weight.grad.clone()
        n_coll = allreduce_gradients(list(net32.parameters()))
        g_avg = net32.mlp_coarse.lin_out.weight.grad.clone()
        q.put((rank, ok, g_local.cpu().numpy(), g_avg.cpu().numpy(), n_coll, bool(d["fine"]["rgb"].requires_grad)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,prec", [("full_ns1", "bf16"), ("full_dtu_ns3", "fp16")])
def test_two_ranks_on_one_card_return_the_unsharded_frame(name, prec):
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU; worker processes must be started before that")
    import torch.multiprocessing as mp
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    os.environ["PYTHONPATH"] = os.pathsep.join([p for p in sys.path if p])           # tests/ and the repo root for the workers
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, prec, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] and res[0][2] > 0


@pytest.mark.parametrize("prec", ["fp16"])
def test_two_ranks_several_objects_and_the_training_call(prec):
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU; worker processes must be started before that")
    import numpy as np
    import torch.multiprocessing as mp
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    os.environ["PYTHONPATH"] = os.pathsep.join([p for p in sys.path if p])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sb2, args=(r, 2, port, prec, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] for r in res), [r[:2] for r in res]          # sharded (2, 333, .) batch == unsharded, bit for bit
    assert all(r[5] for r in res)                               # the training call stayed differentiable (rank-local)
    mean = (res[0][2] + res[1][2]) / 2
    assert np.abs(res[0][2] - res[1][2]).max() > 0              # the ranks really had different batches
    for r in res:
        assert r[4] >= 1 and np.allclose(r[3], mean, rtol=1e-6, atol=1e-9)       # allreduce_gradients = the mean over ranks
