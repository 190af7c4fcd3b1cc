import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x = torch.empty(12800, 3072, dtype=torch.bfloat16, device=dev)
y = torch.empty(12800, 3072, dtype=torch.float32, device=dev)
us = t(lambda: x.fill_(1.0)); print(f"fill bf16 78.6 MB: {us:.1f} us  {x.numel()*2/us/1e6:.2f} TB/s")
us = t(lambda: y.fill_(1.0)); print(f"fill f32 157 MB: {us:.1f} us  {y.numel()*4/us/1e6:.2f} TB/s")
us = t(lambda: x.copy_(y)); print(f"cast f32->bf16 (157 MB read + 78.6 MB write): {us:.1f} us  {(y.numel()*6)/us/1e6:.2f} TB/s")
z = torch.empty(4096, 1024, device=dev)
us = t(lambda: z.fill_(1.0)); print(f"fill f32 16.8 MB: {us:.1f} us  {z.numel()*4/us/1e6:.2f} TB/s")
