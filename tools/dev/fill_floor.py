import torch, time
dev = torch.device("cuda")
for mb in (61, 374):
    n = mb * 1024 * 1024 // 4
    z = torch.empty(n, device=dev)
    for _ in range(5): z.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): z.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"fill_ {mb} MB: {us:.1f} us per launch = {mb*1.048576/us*1e3/1e3:.2f} TB/s", flush=True)
    y = torch.empty_like(z)
    for _ in range(5): y.copy_(z)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50): y.copy_(z)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"copy_ {mb} MB (read+write {2*mb} MB): {us:.1f} us = {2*mb*1.048576/us:.2f} TB/s", flush=True)
