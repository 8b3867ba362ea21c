"""Achievable HBM bandwidth on this box with plain streaming kernels (torch elementwise ops on 2 GiB buffers):
read-only (sum), write-only (fill), copy (read + write).  Context for roofline.frac: the sweeps' counted traffic moves at
~4.9 TB/s."""
import torch
n = 1 << 28                      # 2 GiB of float64
a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
b = torch.empty_like(a)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e-3


gb = n * 8 / 1e9
print(f"read  (sum)   {gb / timed(lambda: a.sum()):8.0f} GB/s")
print(f"write (fill)  {gb / timed(lambda: b.fill_(1.0)):8.0f} GB/s")
print(f"copy  (r+w)   {2 * gb / timed(lambda: b.copy_(a)):8.0f} GB/s")
print(f"axpy  (2r+w)  {3 * gb / timed(lambda: b.add_(a, alpha=2.0)):8.0f} GB/s")
