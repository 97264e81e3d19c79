"""
Host-side helpers around the render path: config access, camera/ray construction and the two tensor
helpers the model protocol uses.  Names follow the reference's src/util/util.py so callers can switch
imports; the implementations are this package's own.
"""
import math

import numpy as np
import torch


class Conf:
    """Uniform read access to a pyhocon ConfigTree (reference configs) or a plain dict."""

    def __init__(self, obj=None):
        self._o = obj if obj is not None else {}

    def _raw(self, key, default):
        o = self._o
        if isinstance(o, Conf):
            return o._raw(key, default)
        try:
            return o[key] if key in o else default
        except TypeError:
            return getattr(o, key, default)

    def get(self, key, default=None):
        return self._raw(key, default)

    def get_int(self, key, default=None):
        return int(self._raw(key, default))

    def get_float(self, key, default=None):
        return float(self._raw(key, default))

    def get_bool(self, key, default=None):
        v = self._raw(key, default)
        return v.lower() in ("1", "true", "yes", "on") if isinstance(v, str) else bool(v)

    def get_string(self, key, default=None):
        return str(self._raw(key, default))

    def get_list(self, key, default=None):
        return self._raw(key, default)

    def __contains__(self, key):
        return self._raw(key, _MISSING) is not _MISSING

    def __getitem__(self, key):
        v = self._raw(key, _MISSING)
        if v is _MISSING:
            raise KeyError(key)
        return Conf(v) if isinstance(v, dict) or hasattr(v, "get_int") else v


_MISSING = object()


def as_conf(c):
    return c if isinstance(c, Conf) else Conf(c)


def repeat_interleave(t, repeats, dim=0):
    """(B, ...) -> (B*repeats, ...) with each row repeated consecutively (reference util.py:58-65)."""
    assert dim == 0
    return t[:, None].expand(t.shape[0], repeats, *t.shape[1:]).reshape(-1, *t.shape[1:])


def combine_interleaved(t, inner_dims=(1,), agg_type="average"):
    """(-1, *inner_dims, C): reduce the first inner dim (views) by mean or max (reference util.py:466-476)."""
    if len(inner_dims) == 1 and inner_dims[0] == 1:
        return t
    t = t.reshape(-1, *inner_dims, *t.shape[1:])
    if agg_type == "average":
        return t.mean(dim=1)
    if agg_type == "max":
        return t.max(dim=1)[0]
    raise NotImplementedError("Unsupported combine type " + agg_type)


def psnr(pred, target):
    mse = float(((pred - target) ** 2).mean())
    return -10.0 * math.log10(mse)


def pose_spherical(theta, phi, radius):
    """c2w (4,4) of a camera at spherical position (theta, phi in degrees) looking at the origin
    (reference util.py:314-328 conventions: camera looks down -z, world z up after the axis flip)."""
    th, ph = math.radians(theta), math.radians(phi)
    t = torch.eye(4)
    t[2, 3] = radius
    rp = torch.eye(4)
    rp[1, 1], rp[1, 2], rp[2, 1], rp[2, 2] = math.cos(ph), -math.sin(ph), math.sin(ph), math.cos(ph)
    rt = torch.eye(4)
    rt[0, 0], rt[0, 2], rt[2, 0], rt[2, 2] = math.cos(th), -math.sin(th), math.sin(th), math.cos(th)
    flip = torch.tensor([[-1.0, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return (flip @ rt @ rp @ t).float()


def unproj_map(width, height, f, c=None, device="cpu"):
    """(H, W, 3) unit camera-space ray directions of a pinhole camera (reference util.py:118-148)."""
    if c is None:
        cx, cy = width * 0.5, height * 0.5
    else:
        c = torch.as_tensor(c).flatten()
        cx, cy = float(c[0]), float(c[1])
    f = torch.as_tensor(f, dtype=torch.float32).flatten()
    fx, fy = float(f[0]), float(f[-1])
    ys = (torch.arange(height, dtype=torch.float32, device=device) - cy) / fy
    xs = (torch.arange(width, dtype=torch.float32, device=device) - cx) / fx
    Y, X = torch.meshgrid(ys, xs, indexing="ij")
    d = torch.stack((X, -Y, -torch.ones_like(X)), dim=-1)
    return d / d.norm(dim=-1, keepdim=True)


def gen_rays(poses, width, height, focal, z_near, z_far, c=None, ndc=False):
    """poses (B,4,4) c2w -> rays (B, H, W, 8) = [origin, direction, near, far] (reference util.py:243-281)."""
    if ndc:
        raise NotImplementedError("NDC rays are not part of the accelerated path")
    dev = poses.device
    dirs = unproj_map(width, height, torch.as_tensor(focal).squeeze(), c=c, device=dev)      # (H,W,3)
    dirs = torch.einsum("bij,hwj->bhwi", poses[:, :3, :3].float(), dirs)
    orig = poses[:, None, None, :3, 3].float().expand(-1, height, width, -1)
    near = torch.full((poses.shape[0], height, width, 1), float(z_near), device=dev)
    far = torch.full((poses.shape[0], height, width, 1), float(z_far), device=dev)
    return torch.cat((orig, dirs, near, far), dim=-1)


class AttrDict(dict):
    """Nested result container with attribute access and toDict(), standing in for dotmap.DotMap in
    the renderer's return value (reference nerf.py:278-316)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def toDict(self):
        return {k: (v.toDict() if isinstance(v, AttrDict) else v) for k, v in self.items()}


def seed_from_torch():
    """A 63-bit seed drawn from torch's global CPU generator, so kernel-side noise follows
    torch.manual_seed like the reference's torch.rand calls do."""
    hi, lo = torch.randint(0, 2 ** 31 - 1, (2,)).tolist()
    return (hi << 31) | lo


def gen_rays_device(pose, width, height, focal, z_near, z_far, c=None, device="cuda"):
    """gen_rays for ONE camera, computed on the GPU by libpnr_hip (pnr_gen_rays): pose (4,4) c2w on the host ->
    rays (H*W, 8) on `device`, never materialised on the host."""
    import ctypes as C
    from . import _native as N
    f = torch.as_tensor(focal, dtype=torch.float32).flatten()
    fx, fy = float(f[0]), float(f[-1])
    cx, cy = (width * 0.5, height * 0.5) if c is None else (float(torch.as_tensor(c).flatten()[0]), float(torch.as_tensor(c).flatten()[1]))
    dev = torch.device(device)
    out = torch.empty(width * height, 8, device=dev, dtype=torch.float32)
    m = (C.c_float * 16)(*[float(v) for v in torch.as_tensor(pose, dtype=torch.float32).cpu().flatten().tolist()])
    N.check(N.lib.pnr_gen_rays(m, int(width), int(height), fx, fy, cx, cy, float(z_near), float(z_far), 0,
                               width * height, N.ptr(out), N.current_stream(dev)), "pnr_gen_rays")
    return out
