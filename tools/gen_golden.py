#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the dev container (needs /root/reference); never shipped
to the GPU box and never imported by the product.

It imports the reference's working hot path UNMODIFIED —
    src/render/nerf.py, src/model/models.py.backup2, encoder.py, resnetfc.py, code.py —
with throw-away stand-ins for the third-party packages that are not installed here
(tools/_shims: dotmap, cv2, pyhocon, torchvision names), feeds it the deterministic inputs of
tests/golden_util.py, records every RNG draw the renderer makes, and writes small .npz fixtures
(inputs + recorded noise + the reference's outputs and a few stage intermediates) into
tests/golden/.  The big tensors (MLP weights, latent maps) are regenerated from the seed by
golden_util, so they are not stored.

    python tools/gen_golden.py            # all cases
    python tools/gen_golden.py tiny_ns1   # some cases
"""
import importlib.machinery
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/src"
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as gu  # noqa: E402


def load_reference():
    import util  # noqa: F401  (reference src/util)
    from render import NeRFRenderer  # reference src/render/nerf.py
    import model  # noqa: F401  reference package (imports the broken models.py; harmless)
    loader = importlib.machinery.SourceFileLoader(
        "model.models_backup2", os.path.join(REF, "model", "models.py.backup2"))
    spec = importlib.util.spec_from_loader("model.models_backup2", loader)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "model"
    sys.modules["model.models_backup2"] = mod
    loader.exec_module(mod)
    return NeRFRenderer, mod.PixelNeRFNet


def model_conf(spec):
    from pyhocon import ConfigFactory
    mlp = dict(type="resnet", n_blocks=spec["n_blocks"], d_hidden=spec["d_hidden"],
               combine_layer=spec["combine_layer"], combine_type=spec["combine_type"])
    conf = dict(
        use_encoder=True, use_global_encoder=False, use_xyz=True, normalize_z=True,
        use_code=True, code=dict(num_freqs=6, freq_factor=1.5, include_input=True),
        use_viewdirs=True, use_code_viewdirs=spec["use_code_viewdirs"],
        mlp_coarse=dict(mlp), mlp_fine=dict(mlp) if spec["fine_mlp"] else dict(type="empty"),
        encoder=dict(backbone="resnet34", pretrained=False, num_layers=4,
                     use_multi_scale=len(spec["lat"]) > 1),
    )
    return ConfigFactory.from_dict(conf)


class NoiseRecorder:
    """Wraps torch.rand_like / rand / randn_like, records the draws in call order
    (reference src/render/nerf.py:111,135,141,158) and optionally plants edge values."""

    def __init__(self, edge):
        self.draws = []
        self.edge = edge
        self._orig = (torch.rand_like, torch.rand, torch.randn_like)

    def __enter__(self):
        o_rl, o_r, o_rn = self._orig

        def rl(t, *a, **k):
            x = o_rl(t, *a, **k)
            if self.edge and x.ndim == 2 and x.shape[0] > 8:
                x[6, 0] = 0.0
                x[7, -1] = float(np.nextafter(np.float32(1), np.float32(0)))
            self.draws.append(("rand_like", x.clone()))
            return x

        def r(*a, **k):
            x = o_r(*a, **k)
            if self.edge and x.ndim == 2 and x.shape[0] > 10:
                x[8, :] = float(np.nextafter(np.float32(1), np.float32(0)))   # u >= cdf[-1] candidates
                x[9, 0] = 0.0
                x[10, :] = 1.0 - 1e-7
            self.draws.append(("rand", x.clone()))
            return x

        def rn(t, *a, **k):
            x = o_rn(t, *a, **k)
            self.draws.append(("randn_like", x.clone()))
            return x

        torch.rand_like, torch.rand, torch.randn_like = rl, r, rn
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand, torch.randn_like = self._orig


def build_case(spec, NeRFRenderer, PixelNeRFNet):
    """The reference's net + renderer for a spec, with seeded weights and injected latents."""
    torch.manual_seed(spec["seed"])
    net = PixelNeRFNet(model_conf(spec))
    net.eval()

    # --- the reference hard-wires latent_size to the ResNet's channel counts (encoder.py:60-73);
    #     for the tiny cases rebuild the reference's own ResnetFC with the synthetic latent width.
    L = gu.d_latent_of(spec)
    if L != net.d_latent:
        from model.resnetfc import ResnetFC
        mk = lambda: ResnetFC(net.d_in, d_out=4, n_blocks=spec["n_blocks"], d_latent=L,
                              d_hidden=spec["d_hidden"], combine_layer=spec["combine_layer"],
                              combine_type=spec["combine_type"])
        net.mlp_coarse = mk()
        if net.mlp_fine is not None:
            net.mlp_fine = mk()
        net.latent_size = L
        net.d_latent = L
    assert net.d_in == gu.d_in_of(spec)

    # --- seeded weights in place of the reference init
    for which, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)):
        if mlp is None:
            continue
        sd = {k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, which).items()}
        missing = set(mlp.state_dict().keys()) ^ set(sd.keys())
        assert not missing, missing
        mlp.load_state_dict(sd, strict=True)

    # --- synthetic latent maps in place of the ResNet trunk
    lats = [torch.from_numpy(x) for x in gu.make_latents(spec)]
    enc = net.encoder
    enc.forward = lambda images: None   # trunk is out of scope; latents are injected
    if len(lats) > 1:
        enc.use_multi_scale = True
        enc.latents = lats
        enc.latent = lats[-1]
    else:
        enc.use_multi_scale = False
        enc.latent = lats[0]
        enc.latents = []

    rays_np, poses_np = gu.make_inputs(spec)
    W, H = spec["image"]
    SB, NS = spec["SB"], spec["NS"]
    images = torch.zeros(SB, NS, 3, H, W)
    net.encode(images, torch.from_numpy(poses_np), torch.tensor(spec["focal"], dtype=torch.float32))

    renderer = NeRFRenderer(
        n_coarse=spec["Kc"], n_fine=spec["Kf"], n_fine_depth=spec["Kfd"],
        depth_std=spec["depth_std"], white_bkgd=spec["white_bkgd"], lindisp=spec["lindisp"])
    renderer.eval()
    return net, renderer, enc, rays_np, poses_np


def run_case(name, spec, NeRFRenderer, PixelNeRFNet):
    net, renderer, enc, rays_np, poses_np = build_case(spec, NeRFRenderer, PixelNeRFNet)

    # --- record stage intermediates
    rec = {"model_calls": [], "index_calls": [], "mlp_calls": []}
    o_fwd = net.forward

    def fwd(xyz, coarse=True, viewdirs=None, far=False):
        out = o_fwd(xyz, coarse=coarse, viewdirs=viewdirs, far=far)
        rec["model_calls"].append((coarse, xyz.clone(), None if viewdirs is None else viewdirs.clone(), out.clone()))
        return out

    net.forward = fwd
    o_index = enc.index

    def index(uv, cam_z=None, image_size=(), z_bounds=None):
        out = o_index(uv, cam_z, image_size, z_bounds)
        rec["index_calls"].append((uv.clone(), out.clone()))
        return out

    enc.index = index
    for mlp in (net.mlp_coarse, net.mlp_fine):
        if mlp is None:
            continue
        o_m = mlp.forward

        def mf(zx, combine_inner_dims=(1,), combine_index=None, dim_size=None, _o=o_m):
            out = _o(zx, combine_inner_dims=combine_inner_dims, combine_index=combine_index, dim_size=dim_size)
            rec["mlp_calls"].append((zx.clone(), out.clone()))
            return out

        mlp.forward = mf

    rays = torch.from_numpy(rays_np)
    with torch.no_grad(), NoiseRecorder(spec["edge"]) as nr:
        out = renderer(net, rays, want_weights=True)

    fx = dict(
        spec_json=np.array(json.dumps(spec)),
        rays=rays_np, poses=poses_np,
        # what encode() left on the module (reference models.py.backup2:121-150)
        enc_w2c=net.poses.numpy().copy(), enc_focal=net.focal.numpy().copy(),
        enc_c=net.c.numpy().copy(), enc_image_shape=net.image_shape.numpy().copy(),
        coarse_rgb=out.coarse.rgb.numpy(), coarse_depth=out.coarse.depth.numpy(),
        coarse_weights=out.coarse.weights.numpy(),
    )
    if renderer.using_fine:
        fx.update(fine_rgb=out.fine.rgb.numpy(), fine_depth=out.fine.depth.numpy(),
                  fine_weights=out.fine.weights.numpy())
    # noise draws, by kind, in the order drawn
    kinds = [k for k, _ in nr.draws]
    fx["noise_order"] = np.array(",".join(kinds))
    for i, (k, x) in enumerate(nr.draws):
        fx[f"noise{i}_{k}"] = x.numpy()
    # model calls: points in / rgb-sigma out (coarse call first, then fine)
    for i, (coarse, xyz, vd, o) in enumerate(rec["model_calls"]):
        tag = "coarse" if coarse else "fine"
        fx[f"pts_xyz_{tag}"] = xyz.numpy()
        fx[f"pts_dirs_{tag}"] = vd.numpy()
        fx[f"pts_out_{tag}"] = o.numpy()
    if spec["d_hidden"] <= 64:   # tiny cases: also keep projection / gather / MLP-input intermediates
        uv, lat = rec["index_calls"][0]
        fx["uv_coarse"] = uv.numpy()
        fx["index_out_coarse"] = lat.numpy()
        zx, mo = rec["mlp_calls"][0]
        fx["mlp_in_coarse"] = zx.numpy()
        fx["mlp_out_coarse"] = mo.numpy()
    os.makedirs(gu.GOLDEN_DIR, exist_ok=True)
    np.savez_compressed(gu.fixture_path(name), **fx)
    sz = os.path.getsize(gu.fixture_path(name))
    print(f"{name}: draws={kinds} coarse_rgb[0]={fx['coarse_rgb'][0, 0]} -> {sz / 1024:.1f} KiB")


def main():
    NeRFRenderer, PixelNeRFNet = load_reference()
    names = sys.argv[1:] or list(gu.CASES)
    for n in names:
        run_case(n, gu.CASES[n], NeRFRenderer, PixelNeRFNet)


if __name__ == "__main__":
    main()
