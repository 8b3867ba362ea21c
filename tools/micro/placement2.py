"""Finer map of tools/micro/placement.py: copy rate (one read + one write stream) and a + b reduction rate (two read streams)
against the distance of the two streams inside ONE allocation.  python tools/micro/placement2.py"""
import torch

n = 1 << 29            # 512 MiB buffers
slab = torch.empty(2 * n + (1 << 30), dtype=torch.uint8, device="cuda")
slab.zero_()
out = torch.empty(n // 8, dtype=torch.float64, device="cuda")


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def rates(d):
    a = slab[0:n].view(torch.float64)
    b = slab[n + d:n + d + n].view(torch.float64)
    tc = timed(lambda: b.copy_(a))
    td = timed(lambda: torch.dot(a, b))
    return 2 * n / tc / 1e12, 2 * n / td / 1e12


def scan(name, ds):
    print(name)
    for d in ds:
        c, r = rates(d)
        print(f"  d = {d:>10d} ({d / (1 << 20):9.4f} MiB): copy {c:.3f}  dot (2 reads) {r:.3f} TB/s", flush=True)


scan("multiples of 1 MiB", [k << 20 for k in range(0, 130)])
scan("multiples of 4 KiB up to 256 KiB", [k << 12 for k in range(0, 65)])
scan("multiples of 64 KiB up to 4 MiB", [k << 16 for k in range(0, 65)])
scan("multiples of 32 MiB up to 1 GiB", [k << 25 for k in range(0, 33)])
