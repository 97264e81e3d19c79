#!/usr/bin/env python3
"""
Golden GRADIENTS.  Dev container only, like gen_golden.py: runs the reference's renderer +
PixelNeRFNet under autograd on the fixture inputs, replaying the noise draws recorded in the forward
fixture, with loss = sum_pass <rgb,G> + <depth,G> + <weights,G> (golden_util.make_loss_weights), and writes
tests/golden/<case>_grad.npz:
    for every MLP parameter of mlp_coarse / mlp_fine and every latent level:
        <key>        the gradient entries at golden_util.grad_sample_index(key, numel) (all entries when small)
        <key>__norm  the l2 norm of the full gradient
    loss             the scalar

    python tools/gen_golden_grad.py [case ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up sys.path for the reference + shims + tests)
import golden_util as gu  # noqa: E402


class NoisePlayer:
    """Replays recorded draws through torch.rand_like / rand / randn_like, in order."""

    def __init__(self, fx):
        kinds = str(fx["noise_order"]).split(",") if str(fx["noise_order"]) else []
        self.q = [(k, torch.from_numpy(fx[f"noise{i}_{k}"])) for i, k in enumerate(kinds)]
        self._orig = (torch.rand_like, torch.rand, torch.randn_like)

    def _pop(self, kind):
        k, x = self.q.pop(0)
        assert k == kind, (k, kind)
        return x.clone()

    def __enter__(self):
        torch.rand_like = lambda t, *a, **k: self._pop("rand_like")
        torch.rand = lambda *a, **k: self._pop("rand")
        torch.randn_like = lambda t, *a, **k: self._pop("randn_like")
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand, torch.randn_like = self._orig
        assert not self.q, "unused recorded draws"


def run(name, NeRFRenderer, PixelNeRFNet):
    fx = gu.load_fixture(name)
    spec = fx["spec"]
    net, renderer, enc, rays_np, _ = gg.build_case(spec, NeRFRenderer, PixelNeRFNet)
    maps = [m.clone().requires_grad_(True) for m in (enc.latents if enc.use_multi_scale else [enc.latent])]
    if enc.use_multi_scale:
        enc.latents, enc.latent = maps, maps[-1]
    else:
        enc.latent = maps[0]
    G = {k: torch.from_numpy(v) for k, v in gu.make_loss_weights(spec).items()}
    with NoisePlayer(fx):
        out = renderer(net, torch.from_numpy(rays_np), want_weights=True)
    loss = 0.0
    for tag in ("coarse", "fine") if renderer.using_fine else ("coarse",):
        lvl = getattr(out, tag)
        loss = loss + (lvl.rgb * G[f"{tag}_rgb"]).sum() + (lvl.depth * G[f"{tag}_depth"]).sum() \
            + (lvl.weights * G[f"{tag}_weights"]).sum()
    # the forward must reproduce the forward fixture exactly (same noise)
    assert np.array_equal(out.coarse.rgb.detach().numpy(), fx["coarse_rgb"])
    loss.backward()
    res = {"loss": np.float64(loss.item())}
    named = []
    for which, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)):
        if mlp is None:
            continue
        for k, p in mlp.named_parameters():
            named.append((f"{which}.{k}", p.grad))
    for i, m in enumerate(maps):
        named.append((f"latent.{i}", m.grad))
    for key, g in named:
        if g is None:       # parameter not reached (e.g. mlp_fine without a fine pass)
            continue
        g = g.detach().numpy().astype(np.float32)
        assert np.isfinite(g).all(), key
        idx = gu.grad_sample_index(key, g.size)
        res[key] = g.reshape(-1)[idx]
        res[key + "__norm"] = np.float64(np.linalg.norm(g.astype(np.float64)))
    path = os.path.join(gu.GOLDEN_DIR, name + "_grad.npz")
    np.savez_compressed(path, **res)
    print(f"{name}: loss={loss.item():.6f} tensors={len(named)} -> {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    NeRFRenderer, PixelNeRFNet = gg.load_reference()
    for n in sys.argv[1:] or gu.GRAD_CASES:
        run(n, NeRFRenderer, PixelNeRFNet)


if __name__ == "__main__":
    main()
