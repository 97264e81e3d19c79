#!/usr/bin/env python3
"""
Golden vectors for SURVEY §8 N1 (ray generation).  Dev container only, like gen_golden.py: imports the reference's
own util.gen_rays / unproj_map / pose_spherical (src/util/util.py:118-148,243-281,314-328) UNMODIFIED (third-party
stand-ins from tools/_shims) and records their outputs for a few cameras into tests/golden/gen_rays.npz:
both principal-point conventions (c=None -> image centre; explicit c), scalar and (fx, fy) focal, non-square and
odd image sizes, a batch of two poses, ndc=False.

    python tools/gen_golden_rays.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden  # noqa: F401,E402  (puts the reference + shims + tests on sys.path)
import golden_util as gu  # noqa: E402

CASES = [
    # name, W, H, focal, c, z_near, z_far, [(theta, phi, radius), ...]
    ("centre_scalar_f", 40, 30, torch.tensor(45.0), None, 1.25, 2.75, [(75.0, -25.0, 2.0)]),
    ("explicit_c_fxfy", 33, 47, torch.tensor([50.0, 60.0]), torch.tensor([15.3, 25.1]), 0.1, 5.0, [(10.0, -20.0, 1.3)]),
    ("one_elem_f_c_1x2", 64, 64, torch.tensor([120.0]), torch.tensor([[31.0, 33.5]]), 1.2, 4.0, [(200.0, -35.0, 2.7)]),
    ("two_poses", 16, 24, torch.tensor(30.0), None, 0.8, 1.8, [(0.0, -20.0, 1.3), (135.0, 10.0, 2.0)]),
]


def main():
    import util  # the reference's src/util
    out = {"names": np.array(",".join(c[0] for c in CASES))}
    for name, W, H, f, c, zn, zf, cams in CASES:
        poses = torch.stack([util.pose_spherical(*cam) for cam in cams])
        rays = util.gen_rays(poses, W, H, f, zn, zf, c=c, ndc=False)
        assert rays.shape == (len(cams), H, W, 8)
        out[f"{name}__poses"] = poses.numpy()
        out[f"{name}__cams"] = np.array(cams, np.float64)
        out[f"{name}__WH"] = np.array([W, H])
        out[f"{name}__focal"] = f.numpy()
        out[f"{name}__c"] = np.zeros(0, np.float32) if c is None else c.numpy()
        out[f"{name}__z"] = np.array([zn, zf], np.float64)
        out[f"{name}__rays"] = rays.numpy()
    path = os.path.join(gu.GOLDEN_DIR, "gen_rays.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
