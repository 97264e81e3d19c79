"""Dev helper (GPU box): error of the MFMA path vs reference fixtures, per stage."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import golden_util as gu
from hip_util import setup

def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 99.0 if mse == 0 else -10 * np.log10(mse)

names = sys.argv[2:] or ["full_ns1", "full_ns3", "full_multiscale_ns2"]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for name in names:
    fx, spec, net, rend = setup(name, precision=prec)
    for tag in ("coarse", "fine"):
        xyz = torch.from_numpy(fx[f"pts_xyz_{tag}"]).cuda(); vd = torch.from_numpy(fx[f"pts_dirs_{tag}"]).cuda()
        out = net(xyz, coarse=(tag == "coarse"), viewdirs=vd).cpu().numpy()
        torch.cuda.synchronize()
        ref = fx[f"pts_out_{tag}"]
        print(f"{name} {prec} pts_{tag}: rgb max|d|={np.abs(out[...,:3]-ref[...,:3]).max():.4f} psnr={psnr(out[...,:3],ref[...,:3]):.1f}dB "
              f"sigma max rel={np.max(np.abs(out[...,3]-ref[...,3])/(1+np.abs(ref[...,3]))):.4f} nan={np.isnan(out).sum()}", flush=True)
    o = rend(net, torch.from_numpy(fx["rays"]).cuda(), want_weights=True)
    torch.cuda.synchronize()
    for lvl in ("coarse", "fine"):
        r = o[lvl].rgb.cpu().numpy(); ref = fx[f"{lvl}_rgb"]
        print(f"{name} {prec} render_{lvl}: rgb max|d|={np.abs(r-ref).max():.4f} psnr={psnr(r,ref):.1f}dB depth max|d|={np.abs(o[lvl].depth.cpu().numpy()-fx[f'{lvl}_depth']).max():.4f}", flush=True)
