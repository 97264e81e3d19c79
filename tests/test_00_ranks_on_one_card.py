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


def _eval_setup():
    import golden_util as gu
    from hip_util import model_conf
    from pixel_nerf_multiscale_amd import NeRFRenderer, PixelNeRFNet
    from test_gpu_eval_loop import _make_dataset
    spec = dict(gu.CASES["full_ns1"])
    torch.manual_seed(0)
    net = PixelNeRFNet(model_conf(spec, "fp32")).cuda().eval()
    for which, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)):
        mlp.load_state_dict({k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, which).items()})
    rend = NeRFRenderer(n_coarse=32, n_fine=16, n_fine_depth=8, white_bkgd=True).cuda().eval()
    data = _make_dataset(net, rend, 2, 4, 32, 32, 33.0, seed=99)
    net.precision = "fp16"
    return net, rend, data


def _worker_eval(rank, world, port, out_dir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pixel_nerf_multiscale_amd import evalio
        net, rend, data = _eval_setup()
        # first call: object 0 only; second call: resume (object 0 skipped on EVERY rank from rank 0's broadcast state)
        m1 = evalio.evaluate(net, rend, data, out_dir, source="0", max_objects=1, verbose=False, seed=99, ray_batch_size=300)
        m2 = evalio.evaluate(net, rend, data, out_dir, source="0", verbose=False, seed=99, ray_batch_size=300)
        q.put((rank, m1, m2))
    finally:
        dist.destroy_process_group()


def test_evaluate_under_two_ranks_has_one_writer_and_the_one_rank_results(tmp_path):
    """evaluate() under a process group: rank 0 alone appends to finish.txt and writes the PNGs, the other
    rank takes the resume state from it, every view is rendered sharded (chunks of 300 rays, each cut over the two ranks)
    and both ranks return the same means — which are the means a one-rank evaluate() of the same seed returns, because
    the jitter is keyed by (seed, view, pixel) whatever the chunking and the number of ranks."""
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU; worker processes must be started before that")
    import torch.multiprocessing as mp
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    os.environ["PYTHONPATH"] = os.pathsep.join([p for p in sys.path if p])
    out = str(tmp_path / "eval2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_eval, args=(r, 2, port, out, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows = [x.split() for x in open(os.path.join(out, "finish.txt")).read().split("\n") if x]
    assert [r[0] for r in rows] == ["obj000", "obj001"]                    # ONE line per object, not one per rank
    # both ranks: the same counts and the same means — up to the ground truth, which every process of THIS test renders for
    # itself (fp32 path behind its own MIOpen trunk: last-bit differences); the evaluated frames are identical on both ranks
    for a, b in ((res[0][1], res[1][1]), (res[0][2], res[1][2])):
        assert a[2] == b[2] and abs(a[0] - b[0]) < 1e-2 and abs(a[1] - b[1]) < 1e-5, (a, b)
    assert res[0][1][2] == 1 and res[0][2][2] == 2
    assert sorted(os.listdir(os.path.join(out, "obj001"))) == ["000001.png", "000002.png", "000003.png"]
    # the one-rank run of the same seed (this process: the workers are gone, the GPU may be initialised now)
    from pixel_nerf_multiscale_amd import evalio
    net, rend, data = _eval_setup()
    out1 = str(tmp_path / "eval1")
    m = evalio.evaluate(net, rend, data, out1, source="0", verbose=False, seed=99)
    assert m[2] == 2
    # the source image goes through the PyTorch / MIOpen trunk in each process (its last bits may differ by algorithm
    # choice): the means agree to that noise; the PNG of a view does too (at most a quantisation step on a few pixels)
    assert m[0] >= 45.0 and abs(m[0] - res[0][2][0]) < 0.5 and abs(m[1] - res[0][2][1]) < 1e-3      # (a 60-dB figure moves by ~0.1 dB with the trunk's last bits)
    a = open(os.path.join(out, "obj001", "000002.png"), "rb").read()
    b = open(os.path.join(out1, "obj001", "000002.png"), "rb").read()
    assert len(a) > 100 and abs(len(a) - len(b)) < 0.2 * len(a)
