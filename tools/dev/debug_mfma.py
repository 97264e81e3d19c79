"""Dev helper (GPU box): isolate MFMA-kernel stages by zeroing parts of the MLP; oracle on CPU is the checker."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import golden_util as gu
from hip_util import build_net
from oracle import pixelnerf_oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else "full_ns1"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
fx = gu.load_fixture(name); spec = fx["spec"]
W, H = spec["image"]
cam = orc.encode_cameras(torch.from_numpy(fx["poses"]), spec["focal"], None, W, H)
lat = [torch.from_numpy(x) for x in gu.make_latents(spec)]
xyz = torch.from_numpy(fx["pts_xyz_coarse"]); vd = torch.from_numpy(fx["pts_dirs_coarse"])

def variant(tag, edit):
    sd = gu.make_mlp_state(spec, "coarse")
    edit(sd)
    net = build_net(spec, fx["poses"], "cuda", prec)
    net.mlp_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net._pack_cache = {}
    out = net(xyz.cuda(), coarse=True, viewdirs=vd.cuda()).cpu()
    ref = orc.point_forward({k: torch.from_numpy(v) for k, v in sd.items()}, cam, lat, xyz, vd, spec["NS"],
                            use_code_viewdirs=spec["use_code_viewdirs"], combine_type=spec["combine_type"])
    d = (out - ref).abs()
    print(f"{tag:28s} rgb max|d|={float(d[...,:3].max()):.5f}  sigma max|d|={float(d[...,3].max()):.4f} (ref sigma max {float(ref[...,3].max()):.2f})  out[0,0]={out[0,0].numpy()} ref[0,0]={ref[0,0].numpy()}", flush=True)

def zero(sd, pats):
    for k in sd:
        if any(p in k for p in pats):
            sd[k][...] = 0

variant("lin_in+lin_out only", lambda sd: zero(sd, ["lin_z", "blocks"]))
variant("+lin_z (no blocks)", lambda sd: zero(sd, ["blocks"]))
variant("+lin_z.0 only", lambda sd: zero(sd, ["blocks", "lin_z.1", "lin_z.2"]))
variant("+lin_z.2 only", lambda sd: zero(sd, ["blocks", "lin_z.0", "lin_z.1"]))
variant("blocks only (no lin_z)", lambda sd: zero(sd, ["lin_z"]))
variant("block0 only", lambda sd: zero(sd, ["lin_z", "blocks.1", "blocks.2", "blocks.3", "blocks.4"]))
variant("block4 only", lambda sd: zero(sd, ["lin_z", "blocks.1", "blocks.2", "blocks.3", "blocks.0"]))
variant("full", lambda sd: None)
