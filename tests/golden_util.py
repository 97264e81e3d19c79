"""
Deterministic synthetic inputs shared by tools/gen_golden.py (which feeds them to the reference,
here in the dev container only) and by the tests (which feed the SAME inputs to the oracle
restatement and to the HIP path).  Everything is drawn from numpy's PCG64 `default_rng(seed)`,
whose streams are stable across numpy versions, so the multi-megabyte weight tensors never have
to be committed: a fixture holds the case spec, the small inputs (rays, poses, recorded noise)
and the reference's outputs.

Nothing here touches /root/reference.
"""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> spec.  lat = list of (C, H, W) per encoder level (one entry = single-scale).
_BASE = dict(
    seed=0, d_hidden=32, n_blocks=5, combine_layer=3, combine_type="average",
    lat=[(8, 5, 7)], NS=1, SB=1, image=(40, 30), focal=45.0, N=16,
    Kc=8, Kf=6, Kfd=2, depth_std=0.01, lindisp=False, white_bkgd=True,
    use_code_viewdirs=False, z_near=1.25, z_far=2.75, radius=2.0, edge=False,
    fine_mlp=True,
)


def _case(**kw):
    d = dict(_BASE)
    d.update(kw)
    return d


CASES = {
    # tiny unit cases (d_hidden 32, L 8)
    "tiny_ns1": _case(seed=11),
    "tiny_ns1_coarse_only": _case(seed=12, Kf=0, Kfd=0),
    "tiny_ns2_lindisp_black": _case(seed=13, NS=2, lindisp=True, white_bkgd=False, z_near=0.4, z_far=4.0),
    "tiny_ns3_edge": _case(seed=14, NS=3, edge=True, N=24),
    "tiny_ns2_codeview": _case(seed=15, NS=2, use_code_viewdirs=True),
    "tiny_multiscale_ns2": _case(seed=16, NS=2, use_code_viewdirs=True, depth_std=1.0,
                                 lat=[(4, 15, 20), (4, 15, 20), (8, 8, 10), (16, 4, 5)]),
    "tiny_sb2_ns2": _case(seed=17, SB=2, NS=2, N=12),
    "tiny_nodepth": _case(seed=18, Kf=6, Kfd=0),
    "tiny_onlydepth": _case(seed=19, Kf=4, Kfd=4),
    "tiny_nofine_mlp": _case(seed=20, fine_mlp=False),
    "tiny_max_combine": _case(seed=21, NS=3, combine_type="max"),
    # full-width cases (d_hidden 512) — what the MFMA kernel is specialised for
    "full_ns1": _case(seed=31, d_hidden=512, lat=[(256, 8, 8)], image=(128, 128), focal=131.25,
                      N=16, Kc=16, Kf=8, Kfd=4),
    "full_ns3": _case(seed=32, d_hidden=512, lat=[(256, 8, 8)], image=(64, 64), focal=120.0, NS=3,
                      N=16, Kc=16, Kf=8, Kfd=4, z_near=1.2, z_far=4.0, radius=2.7),
    "full_multiscale_ns2": _case(seed=33, d_hidden=512, NS=2, use_code_viewdirs=True, depth_std=1.0,
                                 lat=[(64, 16, 16), (64, 16, 16), (128, 8, 8), (256, 4, 4)],
                                 image=(32, 32), focal=33.0, N=16, Kc=16, Kf=8, Kfd=4,
                                 z_near=0.8, z_far=1.8, radius=1.3),
    # BASELINE cfg4 shape (DTU rs_dtu_4): 3 views, 400x300 image, 19x25 latent (T = 475 texels: the general gather path of
    # the fused kernel at NS=3), 128 coarse samples in disparity, black background; edge rays leave the source frusta
    "full_dtu_ns3": _case(seed=34, d_hidden=512, lat=[(256, 19, 25)], image=(400, 300), focal=360.0, NS=3,
                          N=16, Kc=128, Kf=0, Kfd=0, lindisp=True, white_bkgd=False, z_near=0.1, z_far=5.0,
                          radius=2.0, edge=True),
}


# ----------------------------------------------------------------------------- cameras / rays
def pose_spherical(theta_deg, phi_deg, radius):
    """c2w of a camera on a sphere looking at the origin (same convention the reference's
    util.pose_spherical produces, reference src/util/util.py:314-328)."""
    def trans_t(t):
        m = np.eye(4, dtype=np.float32)
        m[2, 3] = t
        return m

    def rot_phi(phi):
        m = np.eye(4, dtype=np.float32)
        m[1, 1] = np.cos(phi); m[1, 2] = -np.sin(phi)
        m[2, 1] = np.sin(phi); m[2, 2] = np.cos(phi)
        return m

    def rot_theta(th):
        m = np.eye(4, dtype=np.float32)
        m[0, 0] = np.cos(th); m[0, 2] = -np.sin(th)
        m[2, 0] = np.sin(th); m[2, 2] = np.cos(th)
        return m

    c2w = trans_t(radius)
    c2w = rot_phi(phi_deg / 180.0 * np.pi) @ c2w
    c2w = rot_theta(theta_deg / 180.0 * np.pi) @ c2w
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    return (flip @ c2w).astype(np.float32)


def pinhole_rays(c2w, W, H, focal, z_near, z_far, pix):
    """Rays [o(3), d(3), near, far] for pixel indices `pix` (flat y*W+x) of a pinhole camera
    looking down -z (reference gen_rays/unproj_map, src/util/util.py:118-148,243-281)."""
    y = (pix // W).astype(np.float32)
    x = (pix % W).astype(np.float32)
    X = (x - W * 0.5) / focal
    Y = (y - H * 0.5) / focal
    d = np.stack([X, -Y, -np.ones_like(X)], -1).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = d @ c2w[:3, :3].T
    o = np.broadcast_to(c2w[:3, 3], d.shape)
    n = np.full((len(pix), 1), z_near, np.float32)
    f = np.full((len(pix), 1), z_far, np.float32)
    return np.concatenate([o, d, n, f], -1).astype(np.float32)


def make_inputs(spec):
    """rays (SB,N,8), poses c2w (SB,NS,4,4), focal (scalar), c (None) for a spec."""
    rng = np.random.default_rng(spec["seed"] * 1000 + 1)
    W, H = spec["image"]
    SB, NS, N = spec["SB"], spec["NS"], spec["N"]
    poses = np.zeros((SB, NS, 4, 4), np.float32)
    rays = np.zeros((SB, N, 8), np.float32)
    for sb in range(SB):
        for v in range(NS):
            poses[sb, v] = pose_spherical(30.0 * v + 11.0 * sb, -20.0, spec["radius"])
        tgt = pose_spherical(75.0 + 5.0 * sb, -25.0, spec["radius"])
        pix = rng.choice(W * H, size=N, replace=False)
        r = pinhole_rays(tgt, W, H, spec["focal"], spec["z_near"], spec["z_far"], pix)
        if spec["edge"]:
            # rays that leave every source frustum / pass behind source cameras / start inside
            r[0, 3:6] = -r[0, 3:6]                       # pointing away from the object
            r[1, 0:3] = poses[sb, 0, :3, 3]              # starts AT source camera 0 (z_cam≈0 hits)
            r[2, 3:6] = np.array([0.0, 0.0, 1.0], np.float32)
            r[3, 3:6] = np.array([1.0, 0.0, 0.0], np.float32)
            r[4, 6] = 0.01                               # very small near
            r[5, 7] = 40.0                               # very large far
        rays[sb] = r
    return rays, poses


# ----------------------------------------------------------------------------- weights / latents
def d_in_of(spec):
    return 78 if spec["use_code_viewdirs"] else 42


def d_latent_of(spec):
    return int(sum(c for c, _, _ in spec["lat"]))


def make_mlp_state(spec, which):
    """ResnetFC state-dict (reference key names, src/model/resnetfc.py:128-158) filled with
    seeded normals.  fc_1 is NOT zero (the reference init zeroes it, which would hide block bugs)."""
    rng = np.random.default_rng(spec["seed"] * 1000 + (2 if which == "coarse" else 3))
    H, L, Din = spec["d_hidden"], d_latent_of(spec), d_in_of(spec)

    def lin(o, i, gain=1.0):
        w = (rng.standard_normal((o, i)) * (gain / np.sqrt(i))).astype(np.float32)
        b = (rng.standard_normal((o,)) * 0.1).astype(np.float32)
        return w, b

    sd = {}
    sd["lin_in.weight"], sd["lin_in.bias"] = lin(H, Din)
    sd["lin_out.weight"], sd["lin_out.bias"] = lin(4, H)
    sd["lin_out.weight"][3] *= 20.0          # make sigma O(10) so transmittance is non-trivial
    sd["lin_out.bias"][3] = sd["lin_out.bias"][3] * 20.0 + 4.0
    for b in range(spec["n_blocks"]):
        sd[f"blocks.{b}.fc_0.weight"], sd[f"blocks.{b}.fc_0.bias"] = lin(H, H, np.sqrt(2.0))
        sd[f"blocks.{b}.fc_1.weight"], sd[f"blocks.{b}.fc_1.bias"] = lin(H, H, 0.7)
    for b in range(min(spec["combine_layer"], spec["n_blocks"])):
        sd[f"lin_z.{b}.weight"], sd[f"lin_z.{b}.bias"] = lin(H, L, 0.7)
    return sd


def make_latents(spec):
    """List (per level) of (SB*NS, C, H, W) fp32 maps = relu(N(0,1)) (post-ReLU ResNet features)."""
    rng = np.random.default_rng(spec["seed"] * 1000 + 4)
    n = spec["SB"] * spec["NS"]
    return [np.maximum(rng.standard_normal((n, c, h, w)), 0).astype(np.float32) for c, h, w in spec["lat"]]


# ----------------------------------------------------------------------------- backward cases
GRAD_CASES = ["tiny_ns1", "tiny_ns2_lindisp_black", "tiny_ns2_codeview", "tiny_multiscale_ns2", "tiny_sb2_ns2",
              "tiny_nodepth", "tiny_onlydepth", "tiny_nofine_mlp", "tiny_max_combine", "tiny_ns1_coarse_only",
              "full_ns1", "full_ns3", "full_multiscale_ns2", "full_dtu_ns3"]
GRAD_SAMPLES = 512      # entries kept per gradient tensor larger than this


def make_loss_weights(spec):
    """Seeded cotangents: loss = sum over passes of <rgb, G_rgb> + <depth, G_depth> + <weights, G_weights>
    (what train/train.py's rgb losses reduce to for one backward, plus depth/weights terms so every output of
    the renderer carries gradient)."""
    rng = np.random.default_rng(spec["seed"] * 1000 + 5)
    SB, N, Kc, Kf = spec["SB"], spec["N"], spec["Kc"], spec["Kf"]
    g = {}
    for tag, K in (("coarse", Kc), ("fine", Kc + Kf)):
        g[f"{tag}_rgb"] = rng.standard_normal((SB, N, 3)).astype(np.float32)
        g[f"{tag}_depth"] = rng.standard_normal((SB, N)).astype(np.float32)
        g[f"{tag}_weights"] = rng.standard_normal((SB, N, K)).astype(np.float32)
    return g


def grad_sample_index(key, numel):
    """Flat indices of the entries of gradient tensor `key` a fixture keeps (all of them when small)."""
    if numel <= GRAD_SAMPLES:
        return np.arange(numel)
    h = int.from_bytes(key.encode(), "little") % (2 ** 31)
    return np.sort(np.random.default_rng(h).choice(numel, size=GRAD_SAMPLES, replace=False))


def load_grad_fixture(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + "_grad.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


# ----------------------------------------------------------------------------- fixtures
def fixture_path(name):
    return os.path.join(GOLDEN_DIR, name + ".npz")


def load_fixture(name):
    z = np.load(fixture_path(name), allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    fx["spec"] = json.loads(str(fx["spec_json"]))
    return fx


def load_rays_fixture():
    """tests/golden/gen_rays.npz (tools/gen_golden_rays.py: the reference's own util.gen_rays outputs): list of dicts."""
    z = np.load(os.path.join(GOLDEN_DIR, "gen_rays.npz"), allow_pickle=False)
    cases = []
    for name in str(z["names"]).split(","):
        c = z[f"{name}__c"]
        cases.append(dict(name=name, poses=z[f"{name}__poses"], cams=z[f"{name}__cams"], W=int(z[f"{name}__WH"][0]),
                          H=int(z[f"{name}__WH"][1]), focal=z[f"{name}__focal"], c=None if c.size == 0 else c,
                          z_near=float(z[f"{name}__z"][0]), z_far=float(z[f"{name}__z"][1]), rays=z[f"{name}__rays"]))
    return cases


def noise_from_fixture(fx):
    """Map the recorded draws (reference draw order, nerf.py:111,135,141,158) to named noise."""
    order = str(fx["noise_order"]).split(",")
    spec = fx["spec"]
    n_imp = spec["Kf"] - spec["Kfd"]
    names = ["noise_c"]
    if spec["Kf"] > 0:
        if n_imp > 0:
            names += ["u", "r"]
        if spec["Kfd"] > 0:
            names += ["g"]
    assert len(names) == len(order), (names, order)
    import torch
    return {n: torch.from_numpy(fx[f"noise{i}_{k}"]) for i, (n, k) in enumerate(zip(names, order))}
