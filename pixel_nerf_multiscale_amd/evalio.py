"""
On-disk / wire formats around the render path, so an evaluation driver built on this package reads
and writes what the reference's eval scripts do.  Pure host-side parsing and arithmetic.

* finish.txt resume log      reference eval/eval.py:113-133,360-362   one line per object: "<name> <psnr> <ssim> <cnt>"
* source-view look-up table   reference eval/eval.py:156-165           viewlist/src_*.txt: "<cat> <obj> <view> [<view> ...]"
* eval view list              reference eval/eval.py:170-176           first line: target view indices
* PNG quantisation            reference eval/eval.py:301               (rgb * 255).astype(uint8) after clamp to [0,1]
* PSNR / SSIM                 reference eval/eval.py:324-332           skimage.measure.compare_psnr / compare_ssim
                              (data_range=1, multichannel): restated from the published definitions — skimage is not
                              installed here and the reference's own numbers need its dataset, so SSIM is "parity unpinned".
* checkpoints                 three mutually incompatible schemas in the reference tree
"""
import os

import numpy as np
import torch


class FinishLog:
    """finish.txt: append-only per-object results with resume (objects already listed are skipped by the driver)."""

    def __init__(self, path):
        self.path = path
        self.finished, self.total_psnr, self.total_ssim, self.cnt = set(), 0.0, 0.0, 0
        if os.path.exists(path):
            with open(path) as f:
                rows = [x.strip().split() for x in f.readlines()]
            rows = [x for x in rows if len(x) == 4]
            self.finished = {x[0] for x in rows}
            self.total_psnr = sum(float(x[1]) for x in rows)
            self.total_ssim = sum(float(x[2]) for x in rows)
            self.cnt = sum(int(x[3]) for x in rows)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self._f = open(path, "a", buffering=1)

    def append(self, obj_name, psnr, ssim, cnt=1):
        self._f.write("{} {} {} {}\n".format(obj_name, psnr, ssim, cnt))
        self.finished.add(obj_name)
        self.total_psnr += psnr
        self.total_ssim += ssim
        self.cnt += cnt

    def mean(self):
        return (self.total_psnr / self.cnt, self.total_ssim / self.cnt) if self.cnt else (0.0, 0.0)

    def close(self):
        self._f.close()


def read_source_view_lut(path):
    """{"<cat>/<obj>": LongTensor(source view indices)}"""
    lut = {}
    with open(path) as f:
        for line in f:
            x = line.strip().split()
            if len(x) >= 3:
                lut[x[0] + "/" + x[1]] = torch.tensor(list(map(int, x[2:])), dtype=torch.long)
    return lut


def read_eval_view_list(path):
    with open(path) as f:
        return torch.tensor(list(map(int, f.readline().split())), dtype=torch.long)


def quantize_uint8(rgb):
    """float image in [0,1] (clamped) -> uint8 the way the reference writes PNGs (truncation, not rounding)."""
    a = np.clip(np.asarray(rgb, dtype=np.float32), 0.0, 1.0)
    return (a * 255).astype(np.uint8)


def psnr(img, gt, data_range=1.0):
    err = np.mean((np.asarray(img, np.float64) - np.asarray(gt, np.float64)) ** 2)
    return 10.0 * np.log10((data_range ** 2) / err)


def ssim(img, gt, data_range=1.0, win_size=7, K1=0.01, K2=0.03):
    """Mean structural similarity, uniform 7x7 window, sample covariance, per channel then averaged, borders cropped
    (Wang et al. 2004 as implemented by skimage.measure.compare_ssim with multichannel=True)."""
    from scipy.ndimage import uniform_filter
    x, y = np.asarray(img, np.float64), np.asarray(gt, np.float64)
    if x.ndim == 3:
        return float(np.mean([ssim(x[..., c], y[..., c], data_range, win_size, K1, K2) for c in range(x.shape[-1])]))
    NP = win_size ** 2
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(x, win_size), uniform_filter(y, win_size)
    uxx, uyy, uxy = uniform_filter(x * x, win_size), uniform_filter(y * y, win_size), uniform_filter(x * y, win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return float(S[pad:-pad, pad:-pad].mean())


def load_checkpoint(net, path, device=None, strict=False):
    """Load any of the reference tree's three checkpoint schemas into a PixelNeRFNet, executing nothing
    from the file (weights_only):  a bare state-dict (models.py.backup2:293-305, upstream pixelNeRF);
    {"net_state_dict": ...} (train/trainlib/trainer.py:593-600);  {"model_state_dict"| "model": ...} (models.py:331-336).
    Returns the (missing, unexpected) key lists of load_state_dict."""
    ck = torch.load(path, map_location=device or "cpu", weights_only=True)
    for key in ("net_state_dict", "model_state_dict", "model", "state_dict"):
        if isinstance(ck, dict) and key in ck and isinstance(ck[key], dict):
            ck = ck[key]
            break
    ck = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ck.items()}   # DataParallel prefix
    res = net.load_state_dict(ck, strict=strict)
    return list(res.missing_keys), list(res.unexpected_keys)
