"""
NeRFRenderer with the reference's Python surface (reference src/render/nerf.py), executing on
libpnr_hip.so:
  * model is this package's PixelNeRFNet  -> ONE native call (pnr_render): sampling, point network,
    compositing, fine resampling all on the GPU, nothing materialised per point in PyTorch;
  * any other callable `model(xyz, coarse=, viewdirs=)` (the reference's protocol, nerf.py:188,212-216)
    -> the HIP sampling / compositing stages around chunked calls of that model.
"""
import ctypes as C

import torch
from torch.autograd.profiler import record_function

from .. import _native as N
from ..util import AttrDict, as_conf, seed_from_torch


class _RenderWrapper(torch.nn.Module):
    """(rays, want_weights) -> (rgb, depth) or nested dict, always with the bound net (nerf.py:15-42)."""

    def __init__(self, net, renderer, simple_output):
        super().__init__()
        self.net, self.renderer, self.simple_output = net, renderer, simple_output

    def forward(self, rays, want_weights=False):
        if rays.shape[0] == 0:
            return torch.zeros(0, 3, device=rays.device), torch.zeros(0, device=rays.device)
        outputs = self.renderer(self.net, rays, want_weights=want_weights and not self.simple_output)
        if self.simple_output:
            lvl = outputs.fine if self.renderer.using_fine else outputs.coarse
            return lvl.rgb, lvl.depth
        return outputs.toDict()


class _ShardedRenderWrapper(torch.nn.Module):
    """The multi-GPU form of _RenderWrapper: same call signature and return structure, every object's rays cut into one
    contiguous range per rank of the process group along dim 1 — as nn.DataParallel(wrapped, gpus, dim=1) of
    nerf.py:367-371 cuts them, any SB — ONE all_gather per call, the full batch returned on every rank
    (parallel.ShardedRenderer).  The render launch writes its pixels (and weights) straight into this rank's slice of the
    gather buffer (pnr_outputs strides): no copies around the collective.
    Differentiable calls (net.train() with autograd on: train/train.py:171,331 calls render_par(all_rays,
    want_weights=True) under the same binding) are NOT sharded: an all_gather is not differentiable and every rank of a
    data-parallel trainer has its own batch, so they run on the local _RenderWrapper path and the step hook is
    parallel.allreduce_gradients(net.parameters()) before optimizer.step()."""

    def __init__(self, net, renderer, simple_output, group=None):
        super().__init__()
        from ..parallel import ShardedRenderer
        self.net, self.renderer, self.simple_output = net, renderer, simple_output
        self.local = _RenderWrapper(net, renderer, simple_output)
        self._want_weights = False
        self.sharded = ShardedRenderer(self._render_shard, group=group, render_into=self._render_into)

    def _levels(self):
        r = self.renderer
        if self.simple_output:
            return ("fine",) if r.using_fine else ("coarse",)
        return ("coarse", "fine") if r.using_fine else ("coarse",)

    def _sample_counts(self):
        r = self.renderer
        return {"coarse": int(r.n_coarse), "fine": int(r.n_coarse) + int(r.n_fine)}

    def _widths(self):
        widths = []
        for lv in self._levels():
            widths += [3, 1] + ([self._sample_counts()[lv]] if self._want_weights else [])
        return widths

    def _keyed(self, base, seed, obj_stride):
        r = self.renderer
        r.ray_index_base, r.forced_seed, r.ray_index_obj_stride = base, seed, obj_stride

    def _render_shard(self, rays, base, seed, obj_stride=0):
        """Fallback route (a renderer without packed outputs): returns the per-level tensors, the caller packs them."""
        self._keyed(base, seed, obj_stride)
        try:
            out = self.renderer(self.net, rays, want_weights=self._want_weights)
        finally:
            self._keyed(0, None, 0)
        cols = []
        for lv in self._levels():
            cols += [out[lv].rgb, out[lv].depth] + ([out[lv].weights] if self._want_weights else [])
        return cols

    def _render_into(self, rays, base, seed, obj_stride, out):
        """rays (SB, n, 8) -> out (SB, n, sum(widths)), this rank's slice of the gather buffer, written by the render launch
        itself.  Returns False when this renderer / model pair has no packed route (the caller then uses _render_shard)."""
        from ..model.models import PixelNeRFNet
        if not (isinstance(self.net, PixelNeRFNet) and hasattr(self.renderer, "_forward_fused") and rays.is_cuda):
            return False
        self._keyed(base, seed, obj_stride)
        try:
            self.renderer.forward_packed(self.net, rays, out, self._levels(), self._want_weights)
        finally:
            self._keyed(0, None, 0)
        return True

    def forward(self, rays, want_weights=False, ray_index_base=0, seed=None):
        """ray_index_base / seed (this package's drivers only): the call's rays are a chunk of a frame rendered under
        `seed`, starting at that ray index — see ShardedRenderer.gather."""
        if rays.shape[0] == 0:
            return torch.zeros(0, 3, device=rays.device), torch.zeros(0, device=rays.device)
        if getattr(self.net, "wants_grad", None) is not None and self.net.wants_grad(rays):
            return self.local(rays, want_weights)          # training: rank-local, see the class docstring
        # per-sample weights (nested output only, nerf.py:33-41) travel in the same all_gather as rgb and depth
        self._want_weights = bool(want_weights) and not self.simple_output
        cols = self.sharded.gather(rays, self._widths(), index_base=ray_index_base, seed=seed)
        if self.simple_output:
            return cols[0], cols[1]
        per = 3 if self._want_weights else 2
        out = {}
        for i, lv in enumerate(self._levels()):
            out[lv] = {"rgb": cols[per * i], "depth": cols[per * i + 1]}
            if self._want_weights:
                out[lv]["weights"] = cols[per * i + 2]
        return out


class NeRFRenderer(torch.nn.Module):
    def __init__(self, n_coarse=128, n_fine=0, n_fine_depth=0, noise_std=0.0, depth_std=0.01,
                 eval_batch_size=100000, white_bkgd=False, lindisp=False, sched=None):
        super().__init__()
        self.n_coarse, self.n_fine, self.n_fine_depth = n_coarse, n_fine, n_fine_depth
        self.noise_std, self.depth_std = noise_std, depth_std
        self.eval_batch_size = eval_batch_size
        self.white_bkgd, self.lindisp = white_bkgd, lindisp
        if lindisp:
            print("Using linear displacement rays")
        self.using_fine = n_fine > 0
        self.sched = sched if sched is not None and len(sched) > 0 else None
        self.register_buffer("iter_idx", torch.tensor(0, dtype=torch.long), persistent=True)
        self.register_buffer("last_sched", torch.tensor(0, dtype=torch.long), persistent=True)
        # Random draws: None -> in-kernel counter-based generator seeded from torch's global generator
        # (so torch.manual_seed governs it, like the reference's torch.rand calls); or a dict with any of
        # noise_c (N,Kc), u (N,Kf-Kfd), r (N,Kf-Kfd), g (N,Kfd) to inject explicit draws (parity tests).
        self.fixed_noise = None
        self.ray_index_base = 0      # global index of rays[0] when a frame is sharded over ranks
        self.ray_index_obj_stride = 0   # rays per object of the UNSHARDED batch when a shard holds a range of several objects' rays
        self.last_seed = None
        self.forced_seed = None      # explicit kernel seed (ShardedRenderer: same seed on every rank)
        self._ws = None

    # ------------------------------------------------------------------ stage wrappers (reference method names)
    def _noise_ptrs(self, dev):
        nz, keep = N.pnr_noise(), []
        if self.fixed_noise:
            for k in ("noise_c", "u", "r", "g"):
                t = self.fixed_noise.get(k)
                if t is not None:
                    t = N.f32c(t, dev)
                    keep.append(t)
                    setattr(nz, k, N.ptr(t))
        return nz, keep

    def _seed(self):
        self.last_seed = seed_from_torch() if self.forced_seed is None else int(self.forced_seed)
        return self.last_seed

    def sample_coarse(self, rays, seed=None):
        """rays (B,8) -> z (B,Kc): stratified, jittered in eval too (nerf.py:98-118)."""
        rays = N.f32c(rays)
        z = torch.empty(rays.shape[0], self.n_coarse, device=rays.device)
        nz, keep = self._noise_ptrs(rays.device)
        N.check(N.lib.pnr_sample_coarse(N.ptr(rays), rays.shape[0], self.n_coarse, int(self.lindisp), nz.noise_c,
                                        self._seed() if seed is None else seed, self.ray_index_base, N.ptr(z),
                                        N.current_stream(rays.device)), "pnr_sample_coarse")
        return z

    def sample_fine_sorted(self, rays, z_coarse, weights, depth, seed=None):
        """sample_fine + sample_fine_depth + cat + sort (nerf.py:120-161,285-295) -> (B, Kc+Kf) ascending."""
        rays = N.f32c(rays)
        B = rays.shape[0]
        z = torch.empty(B, self.n_coarse + self.n_fine, device=rays.device)
        nz, keep = self._noise_ptrs(rays.device)
        w = N.f32c(weights.detach())
        d = N.f32c(depth.detach())
        zc = N.f32c(z_coarse)
        N.check(N.lib.pnr_sample_fine(N.ptr(rays), N.ptr(zc), N.ptr(w), N.ptr(d), B, self.n_coarse, self.n_fine,
                                      self.n_fine_depth, float(self.depth_std), int(self.lindisp), nz.u, nz.r, nz.g,
                                      self.last_seed if seed is None else seed, self.ray_index_base, N.ptr(z),
                                      N.current_stream(rays.device)), "pnr_sample_fine")
        return z

    def composite(self, model, rays, z_samp, coarse=True, sb=0):
        """Generic-model compositing under the reference's profiler label (nerf.py:175)."""
        with record_function("renderer_composite"):
            return self._composite_impl(model, rays, z_samp, coarse, sb)

    def _composite_impl(self, model, rays, z_samp, coarse=True, sb=0):
        """Generic-model compositing (nerf.py:163-249): chunked model calls + the HIP compositing stage."""
        rays, z_samp = N.f32c(rays), N.f32c(z_samp)
        B, K = z_samp.shape
        points = (rays[:, None, :3] + z_samp.unsqueeze(2) * rays[:, None, 3:6])
        use_viewdirs = hasattr(model, "use_viewdirs") and model.use_viewdirs
        if sb > 0:
            points = points.reshape(sb, -1, 3)
            bs, dim = (self.eval_batch_size - 1) // sb + 1, 1
        else:
            points = points.reshape(-1, 3)
            bs, dim = self.eval_batch_size, 0
        if use_viewdirs:
            dirs = rays[:, None, 3:6].expand(-1, K, -1)
            dirs = dirs.reshape(sb, -1, 3) if sb > 0 else dirs.reshape(-1, 3)
            vals = [model(p, coarse=coarse, viewdirs=d)
                    for p, d in zip(torch.split(points, bs, dim=dim), torch.split(dirs, bs, dim=dim))]
        else:
            vals = [model(p, coarse=coarse) for p in torch.split(points, bs, dim=dim)]
        out = N.f32c(torch.cat(vals, dim=dim).reshape(B, K, -1)[..., :4])
        return self._composite_native(rays, z_samp, out)

    def _composite_native(self, rays, z_samp, out):
        B, K = z_samp.shape
        dev = rays.device
        weights = torch.empty(B, K, device=dev)
        rgb = torch.empty(B, 3, device=dev)
        depth = torch.empty(B, device=dev)
        N.check(N.lib.pnr_composite(N.ptr(rays), N.ptr(z_samp), N.ptr(out), B, K, int(bool(self.white_bkgd)),
                                    N.ptr(weights), N.ptr(rgb), N.ptr(depth), N.current_stream(dev)), "pnr_composite")
        return weights, rgb, depth

    # ------------------------------------------------------------------ forward (nerf.py:251-303)
    def forward(self, model, rays, want_weights=False):
        """NeRFRenderer.forward (nerf.py:251-303) under the reference's profiler label (nerf.py:264)."""
        with record_function("renderer_forward"):
            return self._forward_impl(model, rays, want_weights)

    def _forward_impl(self, model, rays, want_weights=False):
        if self.sched is not None and self.last_sched.item() > 0:
            self.n_coarse = self.sched[1][self.last_sched.item() - 1]
            self.n_fine = self.sched[2][self.last_sched.item() - 1]
        assert len(rays.shape) == 3
        if not rays.is_cuda:
            raise RuntimeError("NeRFRenderer runs on the HIP device only: move rays (and the model) to cuda")
        from ..model.models import PixelNeRFNet
        if isinstance(model, PixelNeRFNet):
            if model.wants_grad(rays):
                from .autograd import render_train          # taped fp32 path + explicit backward kernels (N4)
                return render_train(self, model, rays, want_weights)
            if self.training and self.noise_std > 0.0:
                raise NotImplementedError("sigma noise (nerf.py:225-226) exists on the differentiable path only")
            return self._forward_fused(model, rays, want_weights)
        if self.training and self.noise_std > 0.0:
            raise NotImplementedError("sigma noise (nerf.py:225-226) exists on the differentiable path only")
        return self._forward_generic(model, rays, want_weights)

    def forward_packed(self, net, rays, out, levels, want_weights):
        """forward() of a PixelNeRFNet with the outputs of `levels` written as ONE record per ray into out (SB, B, tot):
        per level [rgb(3), depth(1), weights(K) if want_weights] — by the render launch itself (pnr_outputs strides).
        What parallel.ShardedRenderer hands its rank's slice of the all_gather buffer to."""
        if self.sched is not None and self.last_sched.item() > 0:
            self.n_coarse = self.sched[1][self.last_sched.item() - 1]
            self.n_fine = self.sched[2][self.last_sched.item() - 1]
        if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.shape[:2] == rays.shape[:2]):
            raise ValueError("packed output must be a contiguous float32 (SB, B, tot) tensor on the rays' device")
        return self._forward_fused(net, rays, want_weights, packed=(out, tuple(levels)))

    def _forward_fused(self, net, rays, want_weights, camera=None, packed=None):
        """rays (SB, B, 8) through pnr_render; or, with rays=None, the pixels of `camera` = (c2w 16 floats, W, H, fx, fy,
        cx, cy, z_near, z_far, pix0, n) through pnr_render_camera (rays generated inside the render launch).
        packed = (out (SB, B, tot), levels): see forward_packed."""
        if camera is None:
            SB, B, _ = rays.shape
            dev = N.same_device(rays, net.poses)
            rays_f = N.f32c(rays).reshape(-1, 8)
        else:
            SB, B = 1, int(camera[10])
            dev = net.poses.device
        n = SB * B
        Kc, Kf = int(self.n_coarse), int(self.n_fine) if self.using_fine else 0
        # one precision per call: the fused kernel only when BOTH MLPs of this call have a shape it is built for
        prec_c = net.resolved_precision(net.mlp_coarse, net.mlp_fine if Kf > 0 else None)
        prm = net.params_struct(self, prec_c)
        prm.n_fine = Kf
        prm.n_fine_depth = int(self.n_fine_depth) if Kf > 0 else 0
        v, k3 = net.views_struct(prec_c)
        mc, k1 = net.mlp_struct(net.mlp_coarse, prec_c, v)
        fine_mod = net.mlp_fine if (Kf > 0 and net.mlp_fine is not None) else None
        mf, k2 = net.mlp_struct(fine_mod, prec_c, v) if fine_mod is not None else (None, [])
        if v.n_objs != SB:
            raise ValueError(f"rays has {SB} objects but encode() saw {v.n_objs}")
        o = N.pnr_outputs()
        if packed is not None:
            # one record per ray in the caller's buffer: the members of pnr_outputs point INTO it, with the record length as
            # their row stride; levels that are not asked for stay NULL (the library keeps them in its workspace)
            buf, levels = packed
            tot, base, off = buf.shape[-1], buf.data_ptr(), 0
            res = AttrDict()
            for lv in levels:
                if lv == "fine" and Kf == 0:
                    raise ValueError("packed level 'fine' but the renderer has no fine pass")
                K = Kc if lv == "coarse" else Kc + Kf
                res[lv] = AttrDict(rgb=buf[..., off:off + 3], depth=buf[..., off + 3])
                setattr(o, lv + "_rgb", base + 4 * off)
                setattr(o, lv + "_depth", base + 4 * (off + 3))
                off += 4
                if want_weights:
                    res[lv].weights = buf[..., off:off + K]
                    setattr(o, lv + "_weights", base + 4 * off)
                    off += K
            if off != tot:
                raise ValueError(f"packed record of {tot} floats, the levels {levels} need {off}")
            o.rgb_stride = o.depth_stride = o.coarse_weights_stride = o.fine_weights_stride = tot
            if Kf > 0 and "fine" not in levels:         # pnr_render needs somewhere to put the fine pixels
                raise ValueError("a renderer with a fine pass packs its fine level (simple_output picks it)")
        else:
            res = AttrDict(coarse=AttrDict(rgb=torch.empty(SB, B, 3, device=dev), depth=torch.empty(SB, B, device=dev)))
            o.coarse_rgb, o.coarse_depth = N.ptr(res.coarse.rgb), N.ptr(res.coarse.depth)
            if want_weights:
                res.coarse.weights = torch.empty(SB, B, Kc, device=dev)
                o.coarse_weights = N.ptr(res.coarse.weights)
            if Kf > 0:
                res.fine = AttrDict(rgb=torch.empty(SB, B, 3, device=dev), depth=torch.empty(SB, B, device=dev))
                o.fine_rgb, o.fine_depth = N.ptr(res.fine.rgb), N.ptr(res.fine.depth)
                if want_weights:
                    res.fine.weights = torch.empty(SB, B, Kc + Kf, device=dev)
                    o.fine_weights = N.ptr(res.fine.weights)
        if getattr(self, "keep_samples", False) and packed is None:
            res.coarse.z = torch.empty(SB, B, Kc, device=dev)
            o.z_coarse = N.ptr(res.coarse.z)
            if Kf > 0:
                res.fine.z = torch.empty(SB, B, Kc + Kf, device=dev)
                o.z_fine = N.ptr(res.fine.z)
        if getattr(self, "point_events", None) is not None:      # (begin, end) native event handles, see bench.py
            o.ev_point_begin, o.ev_point_end = self.point_events
        nz, k4 = self._noise_ptrs(dev)
        nz.ray_index_obj_stride = int(self.ray_index_obj_stride)
        nbytes = N.lib.pnr_workspace_bytes(C.byref(prm), C.byref(mc), C.byref(v), n)
        if mf is not None:      # the point workspace serves both passes: size it for the larger MLP
            nbytes = max(nbytes, N.lib.pnr_workspace_bytes(C.byref(prm), C.byref(mf), C.byref(v), n))
        ws = net.workspace(nbytes, dev)
        if camera is None:
            N.check(N.lib.pnr_render(C.byref(prm), C.byref(mc), C.byref(mf) if mf is not None else None, C.byref(v),
                                     N.ptr(rays_f), n, B, C.byref(nz), self._seed(), int(self.ray_index_base), C.byref(o),
                                     ws.data_ptr(), ws.numel(), N.current_stream(dev)), "pnr_render")
        else:
            m, W, H, fx, fy, cx, cy, zn, zf, pix0, _ = camera
            N.check(N.lib.pnr_render_camera(C.byref(prm), C.byref(mc), C.byref(mf) if mf is not None else None, C.byref(v),
                                            (C.c_float * 16)(*m), W, H, fx, fy, cx, cy, zn, zf, pix0, n, C.byref(nz),
                                            self._seed(), int(self.ray_index_base), C.byref(o), ws.data_ptr(), ws.numel(),
                                            N.current_stream(dev)), "pnr_render_camera")
        return res

    def _forward_generic(self, model, rays, want_weights):
        if self.ray_index_obj_stride:
            raise NotImplementedError("a shard of several objects' rays is rendered by the PixelNeRFNet path only")
        SB = rays.shape[0]
        r = N.f32c(rays).reshape(-1, 8)
        z_coarse = self.sample_coarse(r)
        wc, rgbc, dc = self.composite(model, r, z_coarse, coarse=True, sb=SB)
        res = AttrDict(coarse=self._format_outputs((wc, rgbc, dc), SB, want_weights))
        if self.using_fine:
            z_all = self.sample_fine_sorted(r, z_coarse, wc, dc)
            res.fine = self._format_outputs(self.composite(model, r, z_all, coarse=False, sb=SB), SB, want_weights)
        return res

    # ------------------------------------------------------------------ whole-frame entry
    def render_image(self, net, pose, width, height, focal, z_near, z_far, c=None):
        """One target view in one call — what the eval drivers do per frame (reference eval/eval.py:250-293:
        gen_rays on the host, H2D, split into ray batches, render_par per chunk with a .cpu() sync each):
        the whole frame is one pnr_render_camera: the rays are generated inside the render launch from the camera
        (no ray tensor, nothing touches the host); bit-identical to util.gen_rays_device + forward.  A model that needs
        the differentiable path gets its rays from pnr_gen_rays and goes through forward.
        pose: (4,4) camera-to-world.  Returns rgb (H, W, 3), depth (H, W) on the device."""
        from .. import util
        from ..model.models import PixelNeRFNet
        dev = net.poses.device
        if not isinstance(net, PixelNeRFNet) or net.num_objs != 1 or net.wants_grad(net.poses):
            rays = util.gen_rays_device(pose, width, height, focal, z_near, z_far, c=c, device=dev)
            out = self(net, rays[None])
        else:
            if self.sched is not None and self.last_sched.item() > 0:
                self.n_coarse = self.sched[1][self.last_sched.item() - 1]
                self.n_fine = self.sched[2][self.last_sched.item() - 1]
            f = torch.as_tensor(focal, dtype=torch.float32).flatten()
            cc = None if c is None else torch.as_tensor(c, dtype=torch.float32).flatten()
            cx, cy = (width * 0.5, height * 0.5) if cc is None else (float(cc[0]), float(cc[1]))
            m = [float(x) for x in torch.as_tensor(pose, dtype=torch.float32).cpu().flatten().tolist()]
            cam = (m, int(width), int(height), float(f[0]), float(f[-1]), cx, cy, float(z_near), float(z_far), 0,
                   int(width) * int(height))
            out = self._forward_fused(net, None, False, camera=cam)
        lvl = out.fine if self.using_fine else out.coarse
        return lvl.rgb.reshape(height, width, 3), lvl.depth.reshape(height, width)

    @staticmethod
    def frame_to_host_async(rgb, depth):
        """Asynchronous D2H of a rendered frame into pinned host buffers (instead of the reference's per-chunk
        blocking .cpu(), eval.py:281-282).  Returns (rgb_host, depth_host, event); event.synchronize() before reading."""
        rgb_h = torch.empty(rgb.shape, dtype=rgb.dtype, pin_memory=True)
        depth_h = torch.empty(depth.shape, dtype=depth.dtype, pin_memory=True)
        rgb_h.copy_(rgb, non_blocking=True)
        depth_h.copy_(depth, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(rgb.device))
        return rgb_h, depth_h, ev

    def _format_outputs(self, rendered, superbatch_size, want_weights=False):
        weights, rgb, depth = rendered
        if superbatch_size > 0:
            rgb = rgb.reshape(superbatch_size, -1, 3)
            depth = depth.reshape(superbatch_size, -1)
            weights = weights.reshape(superbatch_size, -1, weights.shape[-1])
        ret = AttrDict(rgb=rgb, depth=depth)
        if want_weights:
            ret.weights = weights
        return ret

    # ------------------------------------------------------------------ schedule / construction (nerf.py:318-371)
    def sched_step(self, steps=1):
        if self.sched is None:
            return
        self.iter_idx += steps
        while self.last_sched.item() < len(self.sched[0]) and self.iter_idx.item() >= self.sched[0][self.last_sched.item()]:
            self.n_coarse = self.sched[1][self.last_sched.item()]
            self.n_fine = self.sched[2][self.last_sched.item()]
            print("INFO: NeRF sampling resolution changed on schedule ==> c", self.n_coarse, "f", self.n_fine)
            self.last_sched += 1

    @classmethod
    def from_conf(cls, conf, white_bkgd=False, lindisp=False, eval_batch_size=100000):
        conf = as_conf(conf)
        return cls(conf.get_int("n_coarse", 128), conf.get_int("n_fine", 0),
                   n_fine_depth=conf.get_int("n_fine_depth", 0), noise_std=conf.get_float("noise_std", 0.0),
                   depth_std=conf.get_float("depth_std", 0.01), white_bkgd=conf.get_float("white_bkgd", white_bkgd),
                   lindisp=lindisp, eval_batch_size=conf.get_int("eval_batch_size", eval_batch_size),
                   sched=conf.get_list("sched", None))

    def bind_parallel(self, net, gpus=None, simple_output=False):
        """Callable (rays, want_weights) bound to `net` (nerf.py:354-371; callers eval/eval.py:151,
        eval/gen_video.py:110 pass args.gpu_id).  The reference wraps it in nn.DataParallel(dim=1) when several gpus
        are listed; here multi-GPU is one process per GPU: when the script runs under torch.distributed.run (a
        process group exists, world > 1) the returned callable shards every call's rays over the ranks and returns
        the full frame on each (RCCL all_gather over xGMI).  Several ids WITHOUT a process group cannot be served by
        one process: warn and render on this process's device."""
        import torch.distributed as dist
        if gpus is not None and len(gpus) > 1:
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                return _ShardedRenderWrapper(net, self, simple_output=simple_output)
            import warnings
            warnings.warn(f"bind_parallel(gpus={list(gpus)}): single-process multi-GPU (nn.DataParallel) is replaced by one "
                          "process per GPU — start the script with torch.distributed.run to use them; rendering on one "
                          "device now")
        return _RenderWrapper(net, self, simple_output=simple_output)
