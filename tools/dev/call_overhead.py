"""Diagnostic (GPU box): wall time per NeRFRenderer call at small batches vs the kernel time alone (host overhead of the
Python / ctypes layer)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bench
for wl, n in (("srn_chairs_1view_128x128_k64+32", 128), ("srn_chairs_1view_128x128_k64+32", 1024), ("srn_chairs_1view_128x128_k128", 1024)):
    spec, net, rend, rays = bench.build(wl, "bf16", torch.device("cuda"))
    r = rays[:, :n].contiguous()
    for _ in range(20):
        rend(net, r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        rend(net, r)
    t_issue = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 200
    print(f"{wl} rays {n}: host issue {t_issue * 1e6:.0f} us per call, end-to-end {t_total * 1e6:.0f} us per call", flush=True)
