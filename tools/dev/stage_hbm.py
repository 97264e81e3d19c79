"""Diagnostic (GPU box): the stand-alone stage kernels' HBM figures as bench.py reports them."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import bench
for r in bench.stage_kernels_hbm(torch.device("cuda"), n_launch=50):
    print(json.dumps(r), flush=True)
