"""Per-launch kernel time of the headline frame over a long run from a cold start: does the chip ramp up or throttle down?
    python tools/dev/clock_trend.py [n_steps]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
import bench
from pixel_nerf_multiscale_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
spec, net, rend, rays = bench.build(bench.DEFAULT, bench.HEADLINE_DTYPE, dev)
rend(net, rays)          # one launch: packs, workspace
torch.cuda.synchronize()
time.sleep(2.0)          # let the chip idle
evs = []
for _ in range(2 * n):
    h = C.c_void_p(); N.check(N.lib.pnr_event_create(C.byref(h)), "ev"); evs.append(h)
t0 = time.perf_counter()
for i in range(n):
    rend.point_events = (evs[2 * i].value, evs[2 * i + 1].value)
    rend(net, rays)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rend.point_events = None
ks = []
for i in range(n):
    ms = C.c_float(); N.check(N.lib.pnr_event_elapsed_ms(evs[2 * i], evs[2 * i + 1], C.byref(ms)), "el"); ks.append(ms.value)
print(f"{n} launches in {dt*1e3:.1f} ms wall; kernel ms by launch index:")
for a in range(0, n, 10):
    print(f"  {a:4d}-{a+9:4d}: " + " ".join(f"{k:5.2f}" for k in ks[a:a + 10]))
