"""
SpatialEncoder: holder of the pixel-aligned latent map(s) the render kernels sample, plus a plain
torch.nn ResNet trunk (torchvision key names, so upstream checkpoints load) that produces them once per
object via PyTorch-ROCm/MIOpen.  Reference: src/model/encoder.py.  The per-point lookup `index()` of the
reference (encoder.py:138-205) happens inside the HIP point kernel; see csrc/pnr_common.h bilinear_taps.
"""
import torch
from torch import nn

from ..util import as_conf


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _ResNetTrunk(nn.Module):
    """ResNet-18/34 feature trunk with torchvision's module names (conv1, bn1, layer1..4, fc)."""

    def __init__(self, depths):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        chans, cin = (64, 128, 256, 512), 64
        for i, (d, c) in enumerate(zip(depths, chans)):
            blocks = [_BasicBlock(cin if j == 0 else c, c, (2 if i > 0 and j == 0 else 1)) for j in range(d)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
            cin = c
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, 1000)


_DEPTHS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


class SpatialEncoder(nn.Module):
    def __init__(self, backbone="resnet34", pretrained=True, num_layers=4, index_interp="bilinear",
                 index_padding="border", upsample_interp="bilinear", feature_scale=1.0, use_first_pool=True,
                 norm_type="batch", use_multi_scale=False):
        super().__init__()
        if backbone not in _DEPTHS:
            raise NotImplementedError(f"Backbone {backbone} not supported")
        if index_interp != "bilinear" or index_padding != "border":
            raise NotImplementedError("the render kernels implement bilinear/border lookup only")
        self.use_multi_scale, self.num_layers = use_multi_scale, num_layers
        self.feature_scale, self.use_first_pool = feature_scale, use_first_pool
        self.index_interp, self.index_padding, self.upsample_interp = index_interp, index_padding, upsample_interp
        self.align_corners = True
        # pretrained ImageNet weights cannot be downloaded here; load a checkpoint's state-dict instead
        self.model = _ResNetTrunk(_DEPTHS[backbone])
        sizes = [64, 64, 128, 256, 512][:num_layers]
        stem = [self.model.conv1, self.model.bn1, self.model.relu] + ([self.model.maxpool] if use_first_pool else [])
        self.layers = nn.ModuleList([nn.Sequential(*stem)] +
                                    [getattr(self.model, f"layer{i}") for i in range(1, num_layers)])
        self.latent_size = sizes if use_multi_scale else sizes[-1]
        self.latent = None      # plain attributes, like the reference (encoder.py:106-107)
        self.latents = []
        # N2: None = fp32 NCHW trunk (reference numerics).  torch.float16 / torch.bfloat16 = run the trunk under autocast
        # in channels_last; its outputs are then ALREADY the (view, H, W, C) 16-bit images the render kernel gathers
        # from, and are handed over without a repack (pnr_views.latent_packed).
        self.half_dtype = None
        self._level_maps16 = None
        # "latent" = the reference fork's lookup (uv normalised by the latent size: texel coordinate = image-pixel
        # coordinate, image_size ignored — encoder.py:152-164), the default and the parity target; "image" = upstream
        # pixelNeRF's (texel = uv * latent_size / image_size), for checkpoints trained with upstream semantics.  Honoured by
        # index() and by every render / training kernel (pnr_views.uv_scale_x / _y).  Parity unpinned for "image".
        self.uv_scale = "latent"

    def forward(self, x):
        x = x * self.feature_scale
        feats = []
        if self.half_dtype is not None and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
            with torch.autocast("cuda", dtype=self.half_dtype):
                for layer in self.layers:
                    x = layer(x)
                    feats.append(x)
            feats = feats if self.use_multi_scale else [feats[-1]]
            maps16 = [f.to(self.half_dtype).contiguous(memory_format=torch.channels_last) for f in feats]
            self.set_latents([m.float() for m in maps16])
            self._level_maps16 = [m.detach() for m in maps16]
        else:
            for layer in self.layers:
                x = layer(x)
                feats.append(x)
            self.set_latents(feats if self.use_multi_scale else [feats[-1]])
        return self.latents if self.use_multi_scale else self.latent

    def set_latents(self, maps):
        """Install latent map(s) (list of (SB*NS, C, H, W)) — what forward() leaves on the module.  Maps that
        carry an autograd graph (training the trunk, train/train.py:324-346) stay attached to it."""
        keep_graph = torch.is_grad_enabled() and any(m.requires_grad for m in maps)
        maps = [(m if keep_graph else m.detach()).float().contiguous() for m in maps]
        self.latents = maps if self.use_multi_scale else []
        self.latent = maps[-1]
        self._level_maps = maps
        self._level_maps16 = None

    def level_maps(self):
        if self.latent is None:
            raise RuntimeError("encoder has no latent yet: call PixelNeRFNet.encode() first")
        return self.latents if self.use_multi_scale else [self.latent]

    def level_maps16(self, dtype):
        """Channels-last 16-bit maps left by a half-precision forward(), if they match `dtype`; else None."""
        m = self._level_maps16
        if m is None or m[0].dtype != dtype:
            return None
        return m

    def index(self, uv, cam_z=None, image_size=(), z_bounds=None):
        """uv (B, N, 2) image points -> (B, L, N) pixel-aligned features (reference encoder.py:138-205; bilinear, border
        padding, align_corners, every level normalised by its own latent size): a native stage call (pnr_index_latent).
        image_size is ignored like in the reference unless self.uv_scale == "image" (upstream's mapping, opt-in).
        Inference only; inside the render path the lookup is fused into the point kernels."""
        import ctypes as C
        from .. import _native as N
        maps = self.level_maps()
        if not uv.is_cuda or not maps[0].is_cuda:
            raise RuntimeError("SpatialEncoder.index runs on the HIP device only (no CPU / PyTorch evaluation path)")
        nv = maps[0].shape[0]
        if uv.dim() != 3 or uv.shape[-1] != 2 or uv.shape[0] not in (1, nv):
            raise ValueError(f"uv must be (1 or {nv}, N, 2), got {tuple(uv.shape)}")
        dev = N.same_device(uv, maps[0])
        v = N.pnr_views()
        v.n_objs, v.n_views, v.n_levels = nv, 1, len(maps)
        keep = []
        for i, mp in enumerate(maps):
            mp = N.f32c(mp.detach())
            keep.append(mp)
            v.latent[i] = N.ptr(mp)
            v.lat_c[i], v.lat_h[i], v.lat_w[i] = mp.shape[1], mp.shape[2], mp.shape[3]
            if self.uv_scale == "image":
                if len(image_size) != 2:
                    raise ValueError('uv_scale = "image" needs image_size = (W, H)')
                v.uv_scale_x[i], v.uv_scale_y[i] = mp.shape[3] / float(image_size[0]), mp.shape[2] / float(image_size[1])
            elif self.uv_scale != "latent":
                raise ValueError(f"uv_scale must be 'latent' or 'image', got {self.uv_scale!r}")
        q = N.f32c(uv.detach())
        n = q.shape[1]
        out = torch.empty(nv, sum(int(m.shape[1]) for m in maps), n, device=dev)
        N.check(N.lib.pnr_index_latent(C.byref(v), N.ptr(q), n, int(q.shape[0]), N.ptr(out), N.current_stream(dev)),
                "pnr_index_latent")
        return out

    @classmethod
    def from_conf(cls, conf, **kwargs):
        conf = as_conf(conf)
        return cls(backbone=conf.get("backbone", "resnet34"), pretrained=conf.get("pretrained", True),
                   num_layers=conf.get("num_layers", 4), index_interp=conf.get("index_interp", "bilinear"),
                   index_padding=conf.get("index_padding", "border"),
                   upsample_interp=conf.get("upsample_interp", "bilinear"),
                   feature_scale=conf.get("feature_scale", 1.0), use_first_pool=conf.get("use_first_pool", True),
                   norm_type=conf.get("norm_type", "batch"), use_multi_scale=conf.get("use_multi_scale", False),
                   **kwargs)


ImageEncoder = SpatialEncoder
