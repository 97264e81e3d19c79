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

# the sharded call path (parallel.ShardedRenderer at world 1: everything but the collective itself): the render launch
# writes packed (n, 4) records straight into the gather buffer, the outputs are views of it
from pixel_nerf_multiscale_amd.parallel import ShardedRenderer
for wl, n in (("srn_chairs_1view_128x128_k64+32", 2048), ("dtu_3view_400x300_k128", 15000)):
    spec, net, rend, rays = bench.build(wl, "fp16", torch.device("cuda"))
    r = rays[:, :n].contiguous()
    sh = ShardedRenderer.for_model(rend, net, base_seed=1)
    for fn, tag in ((lambda: rend(net, r), "plain forward"), (lambda: sh(r), "ShardedRenderer (packed records, in-place gather buffer)")):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        t_issue = (time.perf_counter() - t0) / 100
        torch.cuda.synchronize()
        t_total = (time.perf_counter() - t0) / 100
        print(f"{wl} rays {n} {tag}: host issue {t_issue * 1e6:.0f} us per call, end-to-end {t_total * 1e6:.0f} us per call", flush=True)
