"""N4: the oracle restatement under torch autograd reproduces the REFERENCE's gradients (fixtures written by
tools/gen_golden_grad.py from the reference's own backward) — pins the backward oracle the GPU tests use."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle_util import noise_from_fixture, oracle_setup
from oracle import pixelnerf_oracle as orc


def oracle_grads(fx):
    """loss and {key: grad} of the oracle render for the fixture's inputs and make_loss_weights cotangents."""
    spec, cam, lat, sd_c, sd_f = oracle_setup(fx)
    lat = [m.clone().requires_grad_(True) for m in lat]
    for sd in (sd_c, sd_f):
        if sd is not None:
            for k in sd:
                sd[k] = sd[k].clone().requires_grad_(True)
    out = orc.render(sd_c, sd_f, cam, lat, torch.from_numpy(fx["rays"]), spec["NS"], spec["Kc"], spec["Kf"], spec["Kfd"],
                     spec["depth_std"], spec["white_bkgd"], spec["lindisp"], noise_from_fixture(fx),
                     use_code_viewdirs=spec["use_code_viewdirs"], n_blocks=spec["n_blocks"],
                     combine_layer=spec["combine_layer"], combine_type=spec["combine_type"])
    G = {k: torch.from_numpy(v) for k, v in gu.make_loss_weights(spec).items()}
    loss = 0.0
    for tag in ("coarse", "fine") if spec["Kf"] > 0 else ("coarse",):
        loss = loss + (out[tag]["rgb"] * G[f"{tag}_rgb"]).sum() + (out[tag]["depth"] * G[f"{tag}_depth"]).sum() \
            + (out[tag]["weights"] * G[f"{tag}_weights"]).sum()
    loss.backward()
    grads = {}
    for which, sd in (("coarse", sd_c), ("fine", sd_f)):
        if sd is not None:
            for k, p in sd.items():
                if p.grad is not None:
                    grads[f"{which}.{k}"] = p.grad.numpy()
    for i, m in enumerate(lat):
        grads[f"latent.{i}"] = m.grad.numpy()
    return float(loss.item()), grads


def compare_grads(got, gfx, rtol, what):
    """Every gradient tensor of the fixture: sampled entries within rtol of the tensor's scale, norm within rtol."""
    keys = [k for k in gfx if k != "loss" and not k.endswith("__norm")]
    assert keys
    for k in keys:
        assert k in got, f"{what}: no gradient for {k}"
        g = np.asarray(got[k], dtype=np.float64).reshape(-1)
        idx = gu.grad_sample_index(k, g.size)
        ref = gfx[k].astype(np.float64)
        norm = float(gfx[k + "__norm"])
        scale = max(norm / np.sqrt(g.size), float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(g[idx] - ref).max())
        assert err <= rtol * scale + 1e-7, f"{what} {k}: max err {err:.3e} vs scale {scale:.3e}"
        assert abs(np.linalg.norm(g) - norm) <= rtol * norm + 1e-7, f"{what} {k}: norm {np.linalg.norm(g)} vs {norm}"


@pytest.mark.parametrize("name", gu.GRAD_CASES)
def test_oracle_autograd_matches_reference_gradients(name):
    fx = gu.load_fixture(name)
    gfx = gu.load_grad_fixture(name)
    loss, grads = oracle_grads(fx)
    assert abs(loss - float(gfx["loss"])) <= 1e-4 * max(1.0, abs(float(gfx["loss"])))
    compare_grads(grads, gfx, 2e-4, name)
