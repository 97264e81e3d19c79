"""Diagnostic (GPU box): section shares of the fused kernel from the -DPNR_STAMPS build."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PNR_LIB", os.path.join(ROOT, "tools", "dev", "libpnr_stamps.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bench
from pixel_nerf_multiscale_amd import _native as N
wl = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT
spec, net, rend, rays = bench.build(wl, os.environ.get("PNR_STAMPS_PRECISION", bench.HEADLINE_DTYPE), torch.device("cuda"))
if len(sys.argv) > 2:          # fewer rays: fewer workgroups active at once (how much of a section is contention between CUs?)
    rays = rays[:, :int(sys.argv[2])].contiguous()
fn = N.lib._cdll.pnr_debug_stamps
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
for _ in range(2): rend(net, rays)
fn(buf, 1)
for _ in range(3): rend(net, rays)
fn(buf, 1)
import time
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5): rend(net, rays)
torch.cuda.synchronize(); print("ms/frame (stamped build)", (time.perf_counter()-t0)/5*1e3)
names = ["tile total", "prologue(geom+posenc) [+ drain of the park stores at the LIN_IN entry]", "gather / taps behind LIN_IN", "lin_z tail of the block loop (tap image / last gather)",
         "park statement (issue only)", "reduce statement", "resblock asm (lin_z prefix + bias + 16 chunks)", "bias + lin_out + store",
         "LIN_IN statement", "compositing of the previous tile's rays", "gather / restore of a lin_z group", "x_stages(8) statement of a lin_z group"]
tot = buf[0]
n_tiles = rays.shape[1] * (spec["Kc"] + (spec["Kc"] + spec["Kf"] if spec["Kf"] else 0)) // 128      # both passes of a coarse+fine render
for i, n in enumerate(names):
    if n != "-":
        print(f"{n:48s} {buf[i]/tot*100:6.2f}%   cycles/wave/tile = {buf[i]/ (3*4*n_tiles):10.0f}")
print(f"in-kernel clock {buf[12]/buf[13]*0.1:.3f} GHz (stamped build)")
