"""Pins the oracle restatement (oracle/pixelnerf_oracle.py) to the reference: every fixture in
tests/golden was produced by running the reference itself (tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import pixelnerf_oracle as orc
from oracle_util import maxdiff, noise_from_fixture, oracle_render, oracle_setup

ALL = sorted(gu.CASES)
TOL = 2e-5   # fp32; the two sides differ only in summation order inside addmm / gather


@pytest.mark.parametrize("name", ALL)
def test_render_matches_reference(name):
    fx = gu.load_fixture(name)
    res = oracle_render(fx)
    spec = fx["spec"]
    span = spec["z_far"] - spec["z_near"]
    for lvl in ("coarse", "fine"):
        if lvl == "fine" and spec["Kf"] == 0:
            continue
        assert maxdiff(res[lvl]["rgb"], fx[f"{lvl}_rgb"]) < TOL, lvl
        assert maxdiff(res[lvl]["weights"], fx[f"{lvl}_weights"]) < TOL, lvl
        assert maxdiff(res[lvl]["depth"], fx[f"{lvl}_depth"]) < TOL * max(span, 1.0) * 4, lvl


@pytest.mark.parametrize("name", ALL)
def test_point_outputs_match_reference(name):
    fx = gu.load_fixture(name)
    spec, cam, lat, sd_c, sd_f = oracle_setup(fx)
    kw = dict(use_code_viewdirs=spec["use_code_viewdirs"], n_blocks=spec["n_blocks"],
              combine_layer=spec["combine_layer"], combine_type=spec["combine_type"])
    out, st = orc.point_forward(sd_c, cam, lat, torch.from_numpy(fx["pts_xyz_coarse"]),
                                torch.from_numpy(fx["pts_dirs_coarse"]), spec["NS"], return_stages=True, **kw)
    assert np.allclose(out.numpy(), fx["pts_out_coarse"], rtol=1e-5, atol=TOL)   # sigma is O(10): relative
    if "uv_coarse" in fx:
        uv_ref = fx["uv_coarse"]
        fin = np.isfinite(uv_ref)
        assert np.allclose(st["uv"].numpy()[fin], uv_ref[fin], rtol=1e-5, atol=1e-4)
        L = st["index_out"].shape[-1]
        ref_idx = np.transpose(fx["index_out_coarse"], (0, 2, 1)).reshape(-1, L)
        assert maxdiff(st["index_out"], ref_idx) < TOL
        assert maxdiff(st["mlp_in"], fx["mlp_in_coarse"]) < 1e-4
        assert maxdiff(st["mlp_out"], fx["mlp_out_coarse"].reshape(-1, 4)) < 1e-4
    if "pts_out_fine" in fx:
        sd = sd_f if sd_f is not None else sd_c
        out = orc.point_forward(sd, cam, lat, torch.from_numpy(fx["pts_xyz_fine"]),
                                torch.from_numpy(fx["pts_dirs_fine"]), spec["NS"], **kw)
        assert np.allclose(out.numpy(), fx["pts_out_fine"], rtol=1e-5, atol=TOL)


@pytest.mark.parametrize("name", ALL)
def test_encode_cameras(name):
    fx = gu.load_fixture(name)
    spec = fx["spec"]
    W, H = spec["image"]
    w2c, focal, c = orc.encode_cameras(torch.from_numpy(fx["poses"]), spec["focal"], None, W, H)
    assert maxdiff(w2c, fx["enc_w2c"]) < 1e-6
    assert maxdiff(focal, fx["enc_focal"]) == 0
    assert maxdiff(c, fx["enc_c"]) == 0
    assert list(fx["enc_image_shape"]) == [W, H]


def test_sample_fine_upper_edge():
    """u >= cdf[-1] gives index Kc (no upper clamp) — planted in the edge fixture."""
    fx = gu.load_fixture("tiny_ns3_edge")
    n = noise_from_fixture(fx)
    assert float(n["u"].max()) >= 0.99999994
    res = oracle_render(fx)
    assert maxdiff(res["fine"]["weights"], fx["fine_weights"]) < TOL


def test_positional_encoding_layout():
    x = torch.tensor([[0.1, -0.2, 0.3]])
    e = orc.positional_encoding(x)
    assert e.shape == (1, 39)
    assert torch.allclose(e[0, :3], x[0])
    assert torch.allclose(e[0, 3:6], torch.sin(1.5 * x[0]))
    assert torch.allclose(e[0, 6:9], torch.sin(1.5 * x[0] + np.float32(np.pi * 0.5)))
    assert torch.allclose(e[0, 9:12], torch.sin(3.0 * x[0]))
