"""
PixelNeRFNet with the reference's Python surface (reference src/model/models.py.backup2, the working
variant — SURVEY.md D3): same constructor / encode / forward / load_weights / save_weights, same
state-dict keys.  forward() and the renderer's fused path evaluate the network in libpnr_hip.so.
"""
import ctypes as C
import os
import os.path as osp
import warnings

import torch
from torch.autograd.profiler import record_function

from .. import _native as N
from ..util import as_conf
from .code import PositionalEncoding
from .encoder import ImageEncoder, SpatialEncoder
from .resnetfc import ResnetFC


def make_encoder(conf, **kwargs):
    conf = as_conf(conf)
    enc_type = conf.get_string("type", "spatial")
    if enc_type in ("spatial", "global"):
        return SpatialEncoder.from_conf(conf, **kwargs)
    raise NotImplementedError("Unsupported encoder type")


def make_mlp(conf, d_in, d_latent=0, allow_empty=False, **kwargs):
    conf = as_conf(conf)
    mlp_type = conf.get_string("type", "mlp")
    if mlp_type == "resnet":
        return ResnetFC.from_conf(conf, d_in, d_latent=d_latent, **kwargs)
    if mlp_type == "empty" and allow_empty:
        return None
    raise NotImplementedError("Unsupported MLP type " + mlp_type)   # 'mlp' (ImplicitNet) is unreachable upstream (D7)


class PixelNeRFNet(torch.nn.Module):
    def __init__(self, conf, stop_encoder_grad=False):
        super().__init__()
        conf = as_conf(conf)
        self.encoder = make_encoder(conf["encoder"])
        self.use_encoder = conf.get_bool("use_encoder", True)
        self.use_xyz = conf.get_bool("use_xyz", False)
        self.normalize_z = conf.get_bool("normalize_z", True)
        self.stop_encoder_grad = stop_encoder_grad
        self.use_code = conf.get_bool("use_code", False)
        self.use_code_viewdirs = conf.get_bool("use_code_viewdirs", True)
        self.use_viewdirs = conf.get_bool("use_viewdirs", False)
        self.use_global_encoder = conf.get_bool("use_global_encoder", False)
        if not (self.use_encoder and self.use_xyz and self.normalize_z and self.use_code and self.use_viewdirs) \
                or self.use_global_encoder:
            raise NotImplementedError(
                "the HIP path implements the shipped configuration family: use_encoder, use_xyz, normalize_z, "
                "use_code, use_viewdirs all true and no global encoder (conf/default.conf)")

        lat = self.encoder.latent_size
        self.latent_size = int(sum(lat)) if isinstance(lat, (list, tuple)) else int(lat)
        d_latent = self.latent_size
        d_in = 6 if self.use_code_viewdirs else 3
        self.code = PositionalEncoding.from_conf(conf["code"], d_in=d_in)
        d_in = self.code.d_out + (0 if self.use_code_viewdirs else 3)
        d_out = 4
        self.mlp_coarse = make_mlp(conf["mlp_coarse"], d_in, d_latent, d_out=d_out)
        self.mlp_fine = make_mlp(conf["mlp_fine"], d_in, d_latent, d_out=d_out, allow_empty=True)
        # world -> camera, bottom row omitted; set by encode()
        self.register_buffer("poses", torch.empty(1, 3, 4), persistent=False)
        self.register_buffer("image_shape", torch.empty(2), persistent=False)
        self.register_buffer("focal", torch.empty(1, 2), persistent=False)
        self.register_buffer("c", torch.empty(1, 2), persistent=False)
        self.d_in, self.d_out, self.d_latent = d_in, d_out, d_latent
        self.num_objs = 0
        self.num_views_per_obj = 1
        # arithmetic of the fc layers: "auto" = fp16 MFMA kernel (fp32 accumulate, saturating activations; same speed as
        # bf16 and ~18 dB closer to the fp32 reference) when the shape allows, else the fp32 HIP path
        self.precision = conf.get_string("precision", os.environ.get("PNR_PRECISION", "auto"))
        self.differentiable = None      # None: follow self.training; True/False: force the taped fp32 path on/off
        # GEMM products of the differentiable path: "fp32" (fp32 MFMA, reference numerics), "bf16x3" (bf16 MFMA on hi/lo
        # operand splits, fp32-class results, 1.5x faster) or "bf16" (plain bf16 products — like the reference's use_amp);
        # fp32 accumulation / tape / gradients in every mode
        self.train_precision = conf.get_string("train_precision", "fp32")
        # tape of the "bf16" mode: "auto" = 16-bit (same gradients, ~0.6 x the bytes), "fp32" = the fp32 tape
        self.train_tape = conf.get_string("train_tape", "auto")
        # one source view with one small latent map: stream W_z . Lat instead of W_z and skip the gather (pnr.h,
        # pnr_pack_mlp_projected); off = always the general gather + lin_z kernel path
        self.project_latent = conf.get_bool("project_latent", os.environ.get("PNR_PROJECT_LATENT", "1") != "0")
        self._pack_cache = {}
        self._struct_cache = {}
        self._ws = None

    # ------------------------------------------------------------------ camera setup (backup2:98-150)
    def encode(self, images, poses, focal, z_bounds=None, c=None):
        self.num_objs = images.size(0)
        if images.dim() == 5:
            assert poses.dim() == 4 and poses.size(1) == images.size(1)
            self.num_views_per_obj = images.size(1)
            images = images.reshape(-1, *images.shape[2:])
            poses = poses.reshape(-1, 4, 4)
        else:
            self.num_views_per_obj = 1
        self.encoder(images)
        self.set_cameras(poses, focal, c, images.shape[-1], images.shape[-2])

    def set_cameras(self, poses, focal, c, width, height):
        """The camera half of encode(): w2c = [R^T | -R^T t], image_shape = (W, H), focal -> (.,2) with fy
        negated, c defaulting to the image centre.  poses: (SB*NS, 4, 4) camera-to-world."""
        dev = self.poses.device
        poses = poses.to(dev).float().reshape(-1, 4, 4)
        rot = poses[:, :3, :3].transpose(1, 2)
        trans = -torch.bmm(rot, poses[:, :3, 3:])
        self.poses = torch.cat((rot, trans), dim=-1).contiguous()
        self.image_shape = torch.tensor([float(width), float(height)], device=dev)
        focal = torch.as_tensor(focal, dtype=torch.float32, device=dev)
        if focal.dim() == 0:
            focal = focal[None, None].repeat(1, 2)
        elif focal.dim() == 1:
            focal = focal.unsqueeze(-1).repeat(1, 2)
        else:
            focal = focal.clone()
        focal = focal.float().clone()
        focal[..., 1] *= -1.0
        self.focal = focal.contiguous()
        if c is None:
            c = (self.image_shape * 0.5).unsqueeze(0)
        else:
            c = torch.as_tensor(c, dtype=torch.float32, device=dev)
            if c.dim() == 0:
                c = c[None, None].repeat(1, 2)
            elif c.dim() == 1:
                c = c.unsqueeze(-1).repeat(1, 2)
        self.c = c.float().contiguous()

    # ------------------------------------------------------------------ native descriptors
    def resolved_precision(self, mlp=None, also=None):
        """Arithmetic of the fc layers for `mlp` (and `also`, the other MLP of the same native call, if any)."""
        p = self.precision
        mlp = mlp if mlp is not None else self.mlp_coarse
        if p == "auto":
            ok = mfma_supported(mlp, self) and (also is None or mfma_supported(also, self))
            return "fp16" if ok else "fp32"
        if p not in N.PRECISIONS:
            raise ValueError(f"unknown precision {p!r}")
        return p

    def params_struct(self, renderer=None, precision=None):
        prm = N.pnr_params()
        if renderer is not None:
            prm.n_coarse, prm.n_fine, prm.n_fine_depth = int(renderer.n_coarse), int(renderer.n_fine), int(renderer.n_fine_depth)
            prm.white_bkgd, prm.lindisp = int(bool(renderer.white_bkgd)), int(bool(renderer.lindisp))
            prm.depth_std = float(renderer.depth_std)
        else:
            prm.n_coarse = 1
        prm.use_code_viewdirs = int(self.use_code_viewdirs)
        prm.num_freqs = int(self.code.num_freqs)
        prm.freq_factor = float(self.code.freq_factor)
        prm.precision = N.PRECISIONS[precision or self.resolved_precision()]
        # several source views in the fused kernel: "16bit" (default) parks the per-view residual streams in the kernel's own
        # 16-bit format until the view reduction; "fp32" parks them as they are (the reference reduces fp32 activations)
        pp = getattr(self, "park_precision", "16bit")
        if pp not in ("16bit", "fp32"):
            raise ValueError(f"park_precision must be '16bit' or 'fp32', got {pp!r}")
        prm.park_fp32 = 1 if pp == "fp32" else 0
        return prm

    def mlp_struct(self, mlp, precision, views=None):
        """pnr_mlp over the module's parameter storage (+ the packed MFMA stream, cached until a
        parameter changes).  With `views` (one object; last latent level 256 channels on <= 256 texels) the stream is the
        projected one (lin_z pre-multiplied with that level's maps, see pnr_pack_mlp_projected) and is also keyed by
        the latent.  Returns (struct, keepalive).  The finished struct is cached too, keyed by everything it points at
        (parameter storages and versions, the projected level), so a steady stream of render calls builds it once."""
        skey = ("mstruct", id(mlp), precision, tuple((p.data_ptr(), p._version) for p in mlp.parameters()),
                self._proj_level_key() if (views is not None and precision != "fp32" and self.project_latent) else None)
        hit = self._struct_cache.get(skey)
        if hit is not None:
            return hit
        self._struct_cache = {k: v for k, v in self._struct_cache.items() if not (k[0] == "mstruct" and k[1] == id(mlp))}
        m, keep = self._build_mlp_struct(mlp, precision, views)
        self._struct_cache[skey] = (m, keep)
        return m, keep

    def _proj_level_key(self):
        mp = self.encoder.level_maps()[-1]
        return (mp.data_ptr(), mp._version, tuple(mp.shape), int(self.num_views_per_obj), len(self.encoder.level_maps()))

    def _build_mlp_struct(self, mlp, precision, views=None):
        if mlp.n_blocks > N.PNR_MAX_BLOCKS:
            raise ValueError(f"ResnetFC with {mlp.n_blocks} blocks: the native library holds at most {N.PNR_MAX_BLOCKS}")
        m = N.pnr_mlp()
        m.d_in, m.d_latent, m.d_hidden, m.d_out = mlp.d_in, mlp.d_latent, mlp.d_hidden, mlp.d_out
        m.n_blocks, m.combine_layer, m.combine_type = mlp.n_blocks, mlp.combine_layer, N.COMBINE[mlp.combine_type]
        keep = []

        def P(t):
            t = N.f32c(t.detach())
            keep.append(t)
            return N.ptr(t)

        m.lin_in_w, m.lin_in_b = P(mlp.lin_in.weight), P(mlp.lin_in.bias)
        m.lin_out_w, m.lin_out_b = P(mlp.lin_out.weight), P(mlp.lin_out.bias)
        for b, blk in enumerate(mlp.blocks):
            m.fc0_w[b], m.fc0_b[b] = P(blk.fc_0.weight), P(blk.fc_0.bias)
            m.fc1_w[b], m.fc1_b[b] = P(blk.fc_1.weight), P(blk.fc_1.bias)
        if mlp.d_latent:
            for b, lz in enumerate(mlp.lin_z):
                m.lin_z_w[b], m.lin_z_b[b] = P(lz.weight), P(lz.bias)
        if precision != "fp32":
            proj_bytes, lat_key = 0, None
            if views is not None and self.project_latent:
                proj_bytes = N.lib.pnr_packed_mlp_projected_bytes(C.byref(m), C.byref(views))
                if proj_bytes:
                    mp = self.encoder.level_maps()[-1]          # the projected (last) level
                    lat_key = (mp.data_ptr(), mp._version, tuple(mp.shape))
                    keep.append(mp)       # the cached struct keeps the map alive: its address cannot be reused under the same key
            key = ("mlp", id(mlp), precision, tuple((p.data_ptr(), p._version) for p in mlp.parameters()), lat_key)
            packed = self._pack_cache.get(key)
            if packed is None:
                self._pack_cache = {k: v for k, v in self._pack_cache.items() if not (k[0] == "mlp" and k[1] == id(mlp))}
                nbytes = proj_bytes or N.lib.pnr_packed_mlp_bytes(C.byref(m))
                if nbytes == 0:
                    raise ValueError("this MLP shape is not supported by the MFMA kernel; use precision='fp32'")
                packed = torch.empty(nbytes, dtype=torch.uint8, device=keep[0].device)
                if proj_bytes:
                    N.check(N.lib.pnr_pack_mlp_projected(C.byref(m), C.byref(views), N.PRECISIONS[precision], packed.data_ptr(),
                                                         nbytes, N.current_stream(packed.device)), "pnr_pack_mlp_projected")
                else:
                    N.check(N.lib.pnr_pack_mlp(C.byref(m), N.PRECISIONS[precision], packed.data_ptr(), nbytes,
                                               N.current_stream(packed.device)), "pnr_pack_mlp")
                self._pack_cache[key] = packed
            m.packed, m.packed_bytes, m.packed_dtype = packed.data_ptr(), packed.numel(), N.PRECISIONS[precision]
            m.packed_texels = int(views.lat_h[views.n_levels - 1] * views.lat_w[views.n_levels - 1]) if proj_bytes else 0
            m.packed_objs = int(views.n_objs) if proj_bytes else 0
            keep.append(packed)
        return m, keep

    def views_struct(self, precision):
        """pnr_views over what encode() left on the module (+ the packed 16-bit maps); cached like mlp_struct."""
        maps = self.encoder.level_maps()
        maps16 = self.encoder.level_maps16(torch.float16 if precision in ("fp16", "f16") else torch.bfloat16) if precision != "fp32" else None
        skey = ("vstruct", precision, int(self.num_views_per_obj), self.uv_scales(),
                tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (self.poses, self.focal, self.c, *maps)),
                None if maps16 is None else tuple(m.data_ptr() for m in maps16))
        hit = self._struct_cache.get(skey)
        if hit is not None:
            return hit
        self._struct_cache = {k: v for k, v in self._struct_cache.items() if k[0] != "vstruct"}
        out = self._build_views_struct(precision)
        self._struct_cache[skey] = out
        return out

    def _build_views_struct(self, precision):
        maps = self.encoder.level_maps()
        dev = maps[0].device
        v, keep = views_from(self.poses, self.focal, self.c, self.num_views_per_obj, maps, self.uv_scales())
        if precision != "fp32":
            tdt = torch.float16 if precision in ("fp16", "f16") else torch.bfloat16
            maps16 = self.encoder.level_maps16(tdt)
            if maps16 is not None:
                # the trunk already produced channels-last 16-bit maps: hand their storage over as is (N2)
                for i, m16 in enumerate(maps16):
                    assert m16.is_contiguous(memory_format=torch.channels_last) and m16.data_ptr() % 16 == 0
                    v.latent_packed[i] = m16.data_ptr()
                    keep.append(m16)
            else:
                key = ("lat", precision, tuple((mp.data_ptr(), mp._version, tuple(mp.shape)) for mp in maps))
                cached = self._pack_cache.get(key)
                if cached is None:
                    self._pack_cache = {k: t for k, t in self._pack_cache.items() if k[0] != "lat"}
                    nbytes = N.lib.pnr_packed_latent_bytes(C.byref(v))
                    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    offs = (C.c_uint64 * N.PNR_MAX_LEVELS)()
                    N.check(N.lib.pnr_pack_latents(C.byref(v), N.PRECISIONS[precision], packed.data_ptr(), nbytes, offs,
                                                   N.current_stream(dev)), "pnr_pack_latents")
                    cached = (packed, list(offs))
                    self._pack_cache[key] = cached
                packed, offs = cached
                for i in range(len(maps)):
                    v.latent_packed[i] = packed.data_ptr() + offs[i]
                keep.append(packed)
            v.packed_dtype = N.PRECISIONS[precision]
        return v, keep

    def uv_scales(self):
        """Per-level (sx, sy) of pnr_views.uv_scale, or None for the fork's mapping.  encoder.uv_scale = "latent" (default):
        the reference fork's SpatialEncoder.index normalises uv by the LATENT size and ignores image_size
        (encoder.py:152-164), so the sampled texel coordinate equals the image-pixel coordinate — the parity
        target.  "image": upstream pixelNeRF's mapping, texel = uv * latent_size / image_size per level — what a checkpoint
        trained with upstream semantics expects.  Parity unpinned: the reference holds no fixture for that mapping."""
        mode = getattr(self.encoder, "uv_scale", "latent")
        if mode == "latent":
            return None
        if mode != "image":
            raise ValueError(f"encoder.uv_scale must be 'latent' or 'image', got {mode!r}")
        W, H = float(self.image_shape[0]), float(self.image_shape[1])
        return tuple((m.shape[3] / W, m.shape[2] / H) for m in self.encoder.level_maps())

    def wants_grad(self, *inputs):
        """True when the call must go through the differentiable (fp32, taped) path: the module is in training
        mode (or `differentiable` is forced True), autograd is on, and a parameter, a latent map or one of `inputs`
        requires grad.  Everything else — eval(), torch.no_grad() — takes the fused inference kernels."""
        if self.differentiable is not None:
            if not self.differentiable:
                return False
        elif not self.training:
            return False
        if not torch.is_grad_enabled():
            return False
        if any(t is not None and t.requires_grad for t in inputs):
            return True
        if any(p.requires_grad for p in self.mlp_coarse.parameters()):
            return True
        if self.mlp_fine is not None and any(p.requires_grad for p in self.mlp_fine.parameters()):
            return True
        return any(m.requires_grad for m in self.latent_maps_for_grad())

    def latent_maps_for_grad(self):
        """Latent maps as autograd sees them: attached to the encoder graph unless stop_encoder_grad
        (models.py.backup2:228-229)."""
        maps = self.encoder.level_maps()
        return [m.detach() for m in maps] if self.stop_encoder_grad else maps

    def workspace(self, nbytes, device):
        """Scratch of a native call, one buffer per (device, stream, host thread): the C ABI is re-entrant (the caller owns
        the workspace), so two streams — or two threads — rendering through ONE net must not be handed the same bytes."""
        import threading
        from collections import OrderedDict
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, int(torch.cuda.current_stream(idx).cuda_stream), threading.get_ident())
        if self._ws is None:
            self._ws = OrderedDict()
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        # least-recently-used bound: short-lived threads / streams would otherwise each leave a full-size buffer behind (a
        # multi-view workspace is hundreds of MB).  An evicted buffer still referenced by a running call stays alive through
        # that reference; callers with more than max_workspaces concurrent streams should raise the bound.
        self._ws.move_to_end(key)
        while len(self._ws) > max(1, int(self.max_workspaces)):
            self._ws.popitem(last=False)
        return ws

    max_workspaces = 8

    def release_workspaces(self):
        """Drop every cached native-call workspace of this net (they are re-created on demand)."""
        self._ws = None

    # ------------------------------------------------------------------ per-point evaluation (backup2:155-282)
    def forward(self, xyz, coarse=True, viewdirs=None, far=False):
        """PixelNeRFNet.forward under the reference's profiler label (models.py.backup2:165)."""
        with record_function("model_inference"):
            return self._forward_impl(xyz, coarse, viewdirs, far)

    def _forward_impl(self, xyz, coarse=True, viewdirs=None, far=False):
        """xyz (SB, B, 3) world points [, viewdirs (SB, B, 3)] -> (SB, B, 4) = sigmoid(rgb), relu(sigma)."""
        assert viewdirs is not None, "use_viewdirs is on: viewdirs required"
        SB, B, _ = xyz.shape
        mlp = self.mlp_coarse if (coarse or self.mlp_fine is None) else self.mlp_fine
        if self.wants_grad(xyz):
            from ..render.autograd import point_mlp_points
            return point_mlp_points(self, mlp, xyz, viewdirs.reshape(SB, B, 3))
        prec = self.resolved_precision(mlp)
        dev = N.same_device(xyz, viewdirs, self.poses)
        xyz_c, vd_c = N.f32c(xyz), N.f32c(viewdirs.reshape(SB, B, 3))
        prm = self.params_struct(None, prec)
        v, k2 = self.views_struct(prec)
        m, k1 = self.mlp_struct(mlp, prec, v)
        if v.n_objs != SB:
            raise ValueError(f"xyz has {SB} objects but encode() saw {v.n_objs}")
        out = torch.empty(SB, B, 4, device=dev, dtype=torch.float32)
        nbytes = N.lib.pnr_workspace_bytes(C.byref(prm), C.byref(m), C.byref(v), 0)
        ws = self.workspace(nbytes, dev)
        N.check(N.lib.pnr_point_mlp(C.byref(prm), C.byref(m), C.byref(v), None, None, 0, N.ptr(xyz_c), N.ptr(vd_c),
                                    SB * B, B, N.ptr(out), ws.data_ptr(), ws.numel(), N.current_stream(dev)),
                "pnr_point_mlp")
        return out

    # ------------------------------------------------------------------ checkpoints (backup2:284-332)
    def load_weights(self, args, opt_init=False, strict=True, device=None):
        if opt_init and not args.resume:
            return
        ckpt_name = "pixel_nerf_init" if opt_init or not args.resume else "pixel_nerf_latest"
        model_path = "%s/%s/%s" % (args.checkpoints_path, args.name, ckpt_name)
        device = self.poses.device if device is None else device
        if os.path.exists(model_path):
            print("Load", model_path)
            self.load_state_dict(torch.load(model_path, map_location=device, weights_only=True), strict=strict)
        elif not opt_init:
            warnings.warn(f"WARNING: {model_path} does not exist, not loaded!! Model will be re-initialized.")
        return self

    def save_weights(self, args, opt_init=False):
        from shutil import copyfile
        ckpt_name = "pixel_nerf_init" if opt_init else "pixel_nerf_latest"
        backup_name = "pixel_nerf_init_backup" if opt_init else "pixel_nerf_backup"
        ckpt_path = osp.join(args.checkpoints_path, args.name, ckpt_name)
        if osp.exists(ckpt_path):
            copyfile(ckpt_path, osp.join(args.checkpoints_path, args.name, backup_name))
        torch.save(self.state_dict(), ckpt_path)
        return self


def views_from(poses, focal, c, num_views_per_obj, maps, uv_scale=None):
    """pnr_views over explicit camera tensors (as encode() leaves them) and fp32 latent maps.  uv_scale: None (the fork's
    texel mapping) or one (sx, sy) per level (PixelNeRFNet.uv_scales)."""
    dev = maps[0].device
    v = N.pnr_views()
    keep = []
    nv = maps[0].shape[0]
    v.n_views = int(num_views_per_obj)
    v.n_objs = nv // v.n_views
    for name, t in (("w2c", poses), ("focal", focal), ("c", c)):
        t = N.f32c(t.detach(), dev)
        keep.append(t)
        setattr(v, name, N.ptr(t))
    if poses.shape[0] != nv:
        raise ValueError(f"{poses.shape[0]} cameras but {nv} latent maps")
    v.n_focal, v.n_c = focal.shape[0], c.shape[0]
    v.n_levels = len(maps)
    for i, mp in enumerate(maps):
        mp = N.f32c(mp.detach())
        keep.append(mp)
        v.latent[i] = N.ptr(mp)
        v.lat_c[i], v.lat_h[i], v.lat_w[i] = mp.shape[1], mp.shape[2], mp.shape[3]
        if uv_scale is not None:
            v.uv_scale_x[i], v.uv_scale_y[i] = float(uv_scale[i][0]), float(uv_scale[i][1])
    return v, keep


def mlp_tensors(mlp):
    """The MLP parameters in a fixed order (autograd inputs of the training Functions)."""
    ts = [mlp.lin_in.weight, mlp.lin_in.bias, mlp.lin_out.weight, mlp.lin_out.bias]
    for blk in mlp.blocks:
        ts += [blk.fc_0.weight, blk.fc_0.bias, blk.fc_1.weight, blk.fc_1.bias]
    if mlp.d_latent:
        for lz in mlp.lin_z:
            ts += [lz.weight, lz.bias]
    return ts


def mfma_supported(mlp, net):
    """Shapes the fused MFMA kernel is specialised for (csrc/point_mfma.hip)."""
    return (mlp.d_hidden == 512 and mlp.d_out == 4 and mlp.d_latent % 256 == 0 and mlp.d_latent > 0
            and mlp.d_in in (42, 78) and net.code.num_freqs == 6 and 1 <= mlp.n_blocks <= 8)
