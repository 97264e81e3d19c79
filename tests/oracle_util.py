"""Glue between fixtures (tests/golden) and the oracle restatement (oracle/)."""
import numpy as np
import torch

import golden_util as gu
from golden_util import noise_from_fixture  # noqa: F401  (fixture plumbing, not oracle code)
from oracle import pixelnerf_oracle as orc


def oracle_setup(fx):
    spec = fx["spec"]
    W, H = spec["image"]
    cam = orc.encode_cameras(torch.from_numpy(fx["poses"]), spec["focal"], None, W, H)
    lat = [torch.from_numpy(x) for x in gu.make_latents(spec)]
    sd_c = {k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, "coarse").items()}
    sd_f = {k: torch.from_numpy(v) for k, v in gu.make_mlp_state(spec, "fine").items()} if spec["fine_mlp"] else None
    return spec, cam, lat, sd_c, sd_f


def oracle_render(fx):
    spec, cam, lat, sd_c, sd_f = oracle_setup(fx)
    return orc.render(sd_c, sd_f, cam, lat, torch.from_numpy(fx["rays"]), spec["NS"], spec["Kc"], spec["Kf"],
                      spec["Kfd"], spec["depth_std"], spec["white_bkgd"], spec["lindisp"],
                      noise_from_fixture(fx), use_code_viewdirs=spec["use_code_viewdirs"],
                      n_blocks=spec["n_blocks"], combine_layer=spec["combine_layer"],
                      combine_type=spec["combine_type"])


def maxdiff(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(a - b)
    d[both_nan] = 0
    return float(np.nanmax(d)) if not np.isnan(d).any() else float("nan")


def oracle_points_f64(fx, tag):
    """The fixture's model call `tag` ('coarse' / 'fine') recomputed by the oracle restatement in float64: the arbiter
    between two fp32 results (the reference's and the HIP path's) that differ by summation order."""
    spec, cam, lat, sd_c, sd_f = oracle_setup(fx)
    sd = sd_c if (tag == "coarse" or sd_f is None) else sd_f
    d = torch.float64
    return orc.point_forward({k: v.to(d) for k, v in sd.items()}, tuple(t.to(d) for t in cam), [m.to(d) for m in lat],
                             torch.from_numpy(fx[f"pts_xyz_{tag}"]).to(d), torch.from_numpy(fx[f"pts_dirs_{tag}"]).to(d),
                             spec["NS"], use_code_viewdirs=spec["use_code_viewdirs"], n_blocks=spec["n_blocks"],
                             combine_layer=spec["combine_layer"], combine_type=spec["combine_type"]).numpy()
