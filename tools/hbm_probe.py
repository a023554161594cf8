import torch, time
dev = torch.device("cuda")
x = torch.randn(768 * 1024 * 1024, device=dev)   # 3 GiB f32
y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
b = x.numel() * 4
ms = t(lambda: x.sum()); print(f"sum f32   {b/ms/1e9:.2f} TB/s read")
ms = t(lambda: x.max()); print(f"max f32   {b/ms/1e9:.2f} TB/s read")
xi = x.view(torch.int32)
ms = t(lambda: xi.sum()); print(f"sum i32   {b/ms/1e9:.2f} TB/s read")
ms = t(lambda: y.copy_(x)); print(f"copy      {2*b/ms/1e9:.2f} TB/s read+write")
ms = t(lambda: y.fill_(1.0)); print(f"fill      {b/ms/1e9:.2f} TB/s write")
xb = x.view(torch.bfloat16)
ms = t(lambda: xb.sum(dtype=torch.float32)); print(f"sum bf16  {b/ms/1e9:.2f} TB/s read")
