"""Measured streaming bandwidth of the box (the 'fraction of achievable' reference of SURVEY 8d): device-to-device
copy (read + write) and a read-only reduction over 8 GiB, timed with HIP events (torch is the measuring tool)."""
import torch
dev = torch.device("cuda:0")
n = 2 * 1024 ** 3  # 8 GiB of f32
x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
y = torch.empty_like(x)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: y.copy_(x))
print(f"copy  : {ms:.3f} ms, {2 * n * 4 / ms / 1e9:.2f} TB/s (read + write)")
ms = timed(lambda: x.sum())
print(f"reduce: {ms:.3f} ms, {n * 4 / ms / 1e9:.2f} TB/s (read only)")
