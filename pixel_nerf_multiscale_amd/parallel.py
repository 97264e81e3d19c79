"""
Multi-GPU rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), replacing the
reference's single-process nn.DataParallel(dim=1) (reference render/nerf.py:367-371, broken there: SURVEY D8).

Rays are independent units, so a batch (SB, B, 8) is cut along B into contiguous ranges, one per rank (every object's
rays alike, as DataParallel(dim=1) does); every rank holds the same packed weights and latents (loaded / encoded
locally), renders its range with the in-kernel noise keyed by the ray's index in the UNSHARDED batch, and ONE collective
per call — an all_gather of (SB * B/world, 4) fp32 [rgb, depth] records that the render launch itself wrote into the
collective's input buffer — reassembles the pixels on every rank.  With the same seed the gathered batch is
bit-identical to the 1-GPU one.  Training is data-parallel instead (one batch per rank, allreduce_gradients below).
"""
import time

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
    """render_shard(rays (SB, n, 8), ray_index_base, seed[, obj_stride]) -> per-ray outputs of its range is the per-rank
    renderer; __call__(rays (SB, B, 8)) returns the full (rgb (SB,B,3), depth (SB,B)) on every rank.  gather() is the
    general form: any list of per-ray outputs, packed into one (SB * rays/world, sum(widths)) fp32 message.
    render_into(rays, base, seed, obj_stride, out (SB, n, tot)) -> bool, when given and returning True, has written the
    packed records straight into `out` — the collective's input buffer — so there is no copy in front of the collective
    (NeRFRenderer.forward_packed: the render launch writes them)."""

    def __init__(self, render_shard, group=None, base_seed=None, render_into=None):
        self.render_shard = render_shard
        self.render_into = render_into
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.frame_idx = 0
        # bench / diagnosis: a list here makes gather() append one (begin, end) pair of events per collective, recorded on the
        # stream the collective is ordered on (device messages), or one host-clock duration in ms (host-memory backends)
        self.collective_timing = None
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
        def keyed(base, seed, obj_stride):
            renderer.ray_index_base, renderer.forced_seed, renderer.ray_index_obj_stride = base, seed, obj_stride

        def render_shard(rays, base, seed, obj_stride=0):
            keyed(base, seed, obj_stride)
            try:
                out = renderer(net, rays)
            finally:
                keyed(0, None, 0)
            lvl = out.fine if renderer.using_fine else out.coarse
            return lvl.rgb, lvl.depth

        def render_into(rays, base, seed, obj_stride, out):
            if not (hasattr(renderer, "forward_packed") and rays.is_cuda and hasattr(net, "wants_grad")):
                return False
            keyed(base, seed, obj_stride)
            try:
                renderer.forward_packed(net, rays, out, ("fine",) if renderer.using_fine else ("coarse",), False)
            finally:
                keyed(0, None, 0)
            return True
        return cls(render_shard, render_into=render_into, **kw)

    def gather(self, rays, widths, index_base=0, seed=None):
        """rays (SB, B, 8), cut along B: rank k renders rays[:, lo_k:hi_k] of EVERY object (nn.DataParallel(dim=1),
        reference render/nerf.py:367-371) with the generator keyed by the rays' indices in the unsharded batch.  ONE
        all_gather of (SB * ceil(B/world), sum(widths)) fp32 per rank, straight from the buffer the render launch wrote;
        returns the full (SB, B, w) / (SB, B) tensors on every rank (views of the gathered buffer when SB == 1).
        index_base / seed: this call's rays are rays [index_base, index_base + B) of a larger frame rendered under `seed`
        (a driver that walks a frame in chunks, eval/eval.py:267-284): the draws then are those of the whole-frame render,
        whatever the chunk size and the number of ranks.  Every rank must pass the same values."""
        assert rays.dim() == 3, "rays (SB, B, 8)"
        SB, B = rays.shape[0], rays.shape[1]
        lo, hi, per = shard_range(B, self.world, self.rank)
        n = hi - lo
        if seed is None:
            seed = frame_seed(self.base_seed, self.frame_idx)
            self.frame_idx += 1
        index_base = int(index_base)
        tot = int(sum(widths))
        obj_stride = B if SB > 1 else 0
        # the rank's own records get their own buffer at world > 1: an all_gather whose input aliases its output is legal for
        # NCCL / RCCL but is the one thing about this path that cannot be rehearsed on a one-card box, and the collective copies
        # the rank's 1/world share either way
        full = torch.empty(self.world, SB * per, tot, device=rays.device, dtype=torch.float32)
        mine = full[0] if self.world == 1 else torch.empty(SB * per, tot, device=rays.device, dtype=torch.float32)
        # (SB * per, tot): rows [0, SB * n) are this rank's records
        if n > 0:
            shard = rays[:, lo:hi] if SB == 1 else rays[:, lo:hi].contiguous()
            slab = mine[: SB * n].view(SB, n, tot)
            if not (self.render_into is not None and self.render_into(shard, index_base + lo, seed, obj_stride, slab)):
                outs = self.render_shard(shard.contiguous(), index_base + lo, seed) if obj_stride == 0 else \
                    self.render_shard(shard.contiguous(), index_base + lo, seed, obj_stride)
                assert len(outs) == len(widths)
                off = 0
                for t, w in zip(outs, widths):
                    slab[..., off:off + w] = t.reshape(SB, n, w)
                    off += w
        if self.world > 1:
            timing = self.collective_timing
            if full.is_cuda and dist.get_backend(self.group) != "nccl":
                # a host-memory backend (gloo) under device tensors — rehearsing the multi-rank path with several ranks on
                # one card, where RCCL refuses duplicate devices: the (small) message travels through the host
                host = torch.empty(self.world, SB * per, tot, dtype=torch.float32)
                mine_h = mine.cpu()
                t0 = time.perf_counter()
                dist.all_gather_into_tensor(host.view(-1), mine_h.view(-1), group=self.group)
                if timing is not None:
                    timing.append((time.perf_counter() - t0) * 1e3)
                full = host.to(rays.device)
            elif full.is_cuda:
                # the collective is ordered behind the render launch on the current stream and the stream waits for it
                # (async_op=False), so a pair of events on that stream brackets exactly the all_gather
                if timing is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(torch.cuda.current_stream(full.device))
                dist.all_gather_into_tensor(full.view(-1), mine.view(-1), group=self.group)
                if timing is not None:
                    e1.record(torch.cuda.current_stream(full.device))
                    timing.append((e0, e1))
            else:
                t0 = time.perf_counter()
                dist.all_gather_into_tensor(full.view(-1), mine.view(-1), group=self.group)
                if timing is not None:
                    timing.append((time.perf_counter() - t0) * 1e3)
        if SB == 1:
            recs = full.view(self.world * per, tot)[:B].unsqueeze(0)                            # (1, B, tot), a view
        else:
            # rank k's slab holds (SB, n_k, tot) in its first SB * n_k rows: stitch the ranges back along B (one copy)
            parts = []
            for k in range(self.world):
                lo_k, hi_k, _ = shard_range(B, self.world, k)
                if hi_k > lo_k:
                    parts.append(full[k][: SB * (hi_k - lo_k)].view(SB, hi_k - lo_k, tot))
            recs = torch.cat(parts, dim=1)
        res, off = [], 0
        for w in widths:
            res.append(recs[..., off:off + w] if w > 1 else recs[..., off])
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
