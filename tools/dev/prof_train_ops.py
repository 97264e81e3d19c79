"""Which PyTorch ops of a training step launch the small copy / fill kernels?  torch.profiler over two steps (GPU box).
    python tools/dev/prof_train_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import golden_util as gu
import hip_util as hu
from torch.profiler import profile, ProfilerActivity

spec = gu._case(seed=5, d_hidden=512, lat=[(256, 8, 8)], image=(128, 128), focal=131.25, NS=1, SB=4, N=128, Kc=64, Kf=32, Kfd=16)
rays_np, poses_np = gu.make_inputs(spec)
net = hu.build_net(spec, poses_np).train()
net.train_precision = "bf16"
maps = [torch.from_numpy(x).cuda().requires_grad_(True) for x in gu.make_latents(spec)]
net.encoder.set_latents(maps)
rend = hu.build_renderer(spec)
rays = torch.from_numpy(rays_np).cuda()
tgt = torch.rand(4, 128, 3, device="cuda")
opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)

def step():
    opt.zero_grad(set_to_none=True)
    out = rend(net, rays, want_weights=True)
    loss = ((out.coarse.rgb - tgt) ** 2).mean() + ((out.fine.rgb - tgt) ** 2).mean()
    loss.backward()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=50, max_src_column_width=110))
