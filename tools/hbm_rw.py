import torch
def t(f, n=20):
    f(); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
for mb in (805, 201, 100, 3200):
    x=torch.empty(mb*1024*1024//2, dtype=torch.bfloat16, device='cuda')
    y=torch.empty_like(x)
    ms=t(lambda: x.fill_(1.0)); print(f"fill {mb} MB: {ms*1e3:.0f} us  {mb*1.048576/ms:.0f} GB/s write")
    ms=t(lambda: y.copy_(x)); print(f"copy {mb} MB: {ms*1e3:.0f} us  {mb*1.048576/ms:.0f} GB/s read + same write")
    ms=t(lambda: x.sum()); print(f"sum  {mb} MB: {ms*1e3:.0f} us  {mb*1.048576/ms:.0f} GB/s read")
