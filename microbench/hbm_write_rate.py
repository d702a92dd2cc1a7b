"""HBM store / load rate of plain torch kernels on this box (context for the tape traffic of the training forward):
fill_ of a 4 GiB tensor (pure stores), sum (pure loads), copy_ (both)."""
import torch
x = torch.empty(1 << 30, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
gb = x.numel() * 4 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f"fill_  {gb / ms:.2f} TB/s  ({ms:.3f} ms for {gb:.1f} GB written)")
ms = t(lambda: x.sum());      print(f"sum    {gb / ms:.2f} TB/s  ({ms:.3f} ms for {gb:.1f} GB read)")
ms = t(lambda: y.copy_(x));   print(f"copy_  {2 * gb / ms:.2f} TB/s  ({ms:.3f} ms for {2 * gb:.1f} GB moved)")
