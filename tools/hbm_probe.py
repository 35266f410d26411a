"""Achievable HBM bandwidth on this GPU with plain torch kernels: device-to-device copy (read + write) and a
read-only reduction, over buffers far larger than the 256 MiB Infinity Cache."""
import torch

dev = "cuda"
n = 1 << 30  # 4 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
y = torch.empty_like(x)


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


t = timed(lambda: y.copy_(x))
print(f"copy 4 GiB -> 4 GiB : {2 * n * 4 / t / 1e12:5.2f} TB/s (read + write)")
t = timed(lambda: torch.sum(x))
print(f"sum over 4 GiB      : {n * 4 / t / 1e12:5.2f} TB/s (read)")
xb = x.view(torch.bfloat16)
t = timed(lambda: torch.amax(xb))
print(f"amax over 4 GiB bf16: {n * 4 / t / 1e12:5.2f} TB/s (read)")
print(torch.cuda.get_device_properties(0))
