"""What plain streaming kernels reach on this GPU (GPU box): torch fill / zero (writes), copy (read + write)
and sum (read) over 2 GiB.  Reference points for the tape writers and readers of the training step."""
import torch, time
x = torch.empty(2**29, dtype=torch.float32, device="cuda")   # 2 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
gb = x.numel() * 4 / 1e9
print("fill  TB/s", gb / t(lambda: x.fill_(1.0)) / 1e3)
print("zero  TB/s", gb / t(lambda: x.zero_()) / 1e3)
print("copy  TB/s (r+w)", 2 * gb / t(lambda: y.copy_(x)) / 1e3)
print("sum   TB/s (read)", gb / t(lambda: x.sum()) / 1e3)
