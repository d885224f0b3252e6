import torch, time
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N = 1233125376  # 3.2M x 384
x = torch.empty(N, dtype=torch.bfloat16, device="cuda")
y = torch.empty(N, dtype=torch.bfloat16, device="cuda")
ms = t(lambda: x.fill_(1.0)); print(f"fill  {2*N/ms/1e6:.0f} GB/s  {ms*1e3:.0f} us")
ms = t(lambda: y.copy_(x)); print(f"copy  {4*N/ms/1e6:.0f} GB/s (r+w)  {ms*1e3:.0f} us")
ms = t(lambda: x.sum()); print(f"read(sum)  {2*N/ms/1e6:.0f} GB/s  {ms*1e3:.0f} us")
z = torch.empty(N // 2, dtype=torch.float32, device="cuda")
ms = t(lambda: z.fill_(1.0)); print(f"fill f32  {2*N/ms/1e6:.0f} GB/s  {ms*1e3:.0f} us")
