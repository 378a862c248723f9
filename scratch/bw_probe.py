import torch, time
dev = torch.device('cuda', 0)
n = 1_280_000_000 // 4
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.sum()); print(f"sum of 1.28 GB: {ms:.3f} ms = {1.28/ms:.2f} TB/s read")
ms = t(lambda: y.copy_(x)); print(f"copy 1.28 GB: {ms:.3f} ms = {2.56/ms:.2f} TB/s read+write")
ms = t(lambda: y.fill_(1.0)); print(f"fill 1.28 GB: {ms:.3f} ms = {1.28/ms:.2f} TB/s write")
z = torch.empty(n // 4, device=dev)
ms = t(lambda: z.copy_(x[: n // 4])); print(f"copy 0.32 GB: {ms:.3f} ms = {0.64/ms:.2f} TB/s")
