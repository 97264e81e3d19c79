"""
Multi-GPU rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), replacing the
reference's single-process nn.DataParallel(dim=1) (reference render/nerf.py:367-371, broken there: SURVEY D8).

Rays are independent units, so a frame is cut into contiguous ray ranges, one per rank; every rank holds the
same packed weights and latents (loaded / encoded locally), renders its range with the in-kernel noise keyed
by the GLOBAL ray index, and ONE collective per frame — an all_gather of (rays/world, 4) fp32 [rgb, depth] —
reassembles the pixels on every rank.  With the same seed the gathered frame is bit-identical to the 1-GPU frame.
"""
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """Contiguous [lo, hi) of rank's rays; every rank gets ceil(n/world) except the tail."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


def frame_seed(base_seed, frame_idx):
    """Per-frame kernel seed derived without communication (splitmix64 step)."""
    z = (int(base_seed) + 0x9E3779B97F4A7C15 * (int(frame_idx) + 1)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return (z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF


class ShardedRenderer:
    """render_shard(rays (1, n, 8), ray_index_base, seed) -> (rgb (1,n,3), depth (1,n)) is the per-rank renderer;
    __call__(rays (1, B, 8)) returns the full (rgb (1,B,3), depth (1,B)) on every rank.  gather() is the general
    form: any list of per-ray outputs, packed into one (rays/world, sum(widths)) fp32 message."""

    def __init__(self, render_shard, group=None, base_seed=None):
        self.render_shard = render_shard
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.frame_idx = 0
        if base_seed is None:
            t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
            if self.world > 1:
                dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
                t = t.to(dev)
                dist.broadcast(t, src=0, group=group)      # once, at construction
            base_seed = int(t.item())
        self.base_seed = base_seed

    @classmethod
    def for_model(cls, renderer, net, **kw):
        """Bind a NeRFRenderer + PixelNeRFNet (simple_output semantics: fine if using_fine else coarse)."""
        def render_shard(rays, base, seed):
            renderer.ray_index_base, renderer.forced_seed = base, seed
            try:
                out = renderer(net, rays)
            finally:
                renderer.ray_index_base, renderer.forced_seed = 0, None
            lvl = out.fine if renderer.using_fine else out.coarse
            return lvl.rgb, lvl.depth
        return cls(render_shard, **kw)

    def gather(self, rays, widths):
        """rays (1, B, 8); render_shard returns len(widths) tensors of (1, n, w) / (1, n) for its range.  ONE
        all_gather of (ceil(B/world), sum(widths)) fp32; returns the full (1, B, w) / (1, B) tensors on every rank."""
        assert rays.dim() == 3 and rays.shape[0] == 1, "sharded rendering takes one object per call: rays (1, B, 8)"
        B = rays.shape[1]
        lo, hi, per = shard_range(B, self.world, self.rank)
        seed = frame_seed(self.base_seed, self.frame_idx)
        self.frame_idx += 1
        tot = int(sum(widths))
        pix = torch.zeros(per, tot, device=rays.device, dtype=torch.float32)
        if hi > lo:
            outs = self.render_shard(rays[:, lo:hi].contiguous(), lo, seed)
            assert len(outs) == len(widths)
            off = 0
            for t, w in zip(outs, widths):
                pix[: hi - lo, off:off + w] = t.reshape(hi - lo, w)
                off += w
        if self.world > 1:
            if pix.is_cuda and dist.get_backend(self.group) != "nccl":
                # a host-memory backend (gloo) under device tensors — rehearsing the multi-rank path with several ranks on
                # one card, where RCCL refuses duplicate devices: the (small) message travels through the host
                host = torch.empty(self.world * per, tot, dtype=torch.float32)
                dist.all_gather_into_tensor(host, pix.cpu(), group=self.group)
                full = host.to(rays.device)
            else:
                full = torch.empty(self.world * per, tot, device=rays.device, dtype=torch.float32)
                dist.all_gather_into_tensor(full, pix, group=self.group)
        else:
            full = pix
        full = full[:B]
        res, off = [], 0
        for w in widths:
            res.append(full[:, off:off + w].reshape(1, B, w) if w > 1 else full[:, off].reshape(1, B))
            off += w
        return res

    def __call__(self, rays):
        rgb, depth = self.gather(rays, [3, 1])
        return rgb, depth


# ------------------------------------------------------------------------------------------------------------
# Data-parallel training (SURVEY §8 N4 + e): every rank renders + back-propagates its own ray batch through the
# HIP backward kernels (render/autograd.py); gradients are averaged with ONE all-reduce per bucket over RCCL.
# The reference has no multi-GPU training path at all (its DataParallel wrapper is eval-only and broken, D8).
def allreduce_gradients(params, group=None, bucket_bytes=64 << 20, average=True):
    """Average .grad of `params` over the ranks of `group`: gradients are packed into flat fp32 buckets of at most
    bucket_bytes (xGMI rings are per-link bound: few large messages beat many small ones), all-reduced in place and
    copied back.  Parameters without a gradient on this rank contribute zeros (every rank must see the same list).
    Returns the number of collectives issued."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_coll, i = 0, 0
    while i < len(params):
        j, n = i, 0
        while j < len(params) and (j == i or (n + params[j].numel()) * 4 <= bucket_bytes):
            n += params[j].numel()
            j += 1
        dev = params[i].device
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params[i:j]:
            if p.grad is not None:
                flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
            off += p.numel()
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            n_coll += 1
            if average:
                flat.div_(world)
        off = 0
        for p in params[i:j]:
            g = flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += p.numel()
        i = j
    return n_coll
