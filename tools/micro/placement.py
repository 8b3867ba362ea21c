"""HBM copy rate of one MI355X against the DISTANCE between source and destination inside one allocation (HISTORY R5.18: the
launch time of the level-0 sweeps depends on where their arrays are; does a plain copy see it, and with which period?).
python tools/micro/placement.py [GiB of the buffers]"""
import sys
import torch

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(gib * (1 << 30))
slab = torch.empty(3 * n + (1 << 30), dtype=torch.uint8, device="cuda")
slab.zero_()


def rate(off, reps=12):
    src = slab[0:n].view(torch.float64)
    dst = slab[off:off + n].view(torch.float64)
    for _ in range(3):
        dst.copy_(src)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        dst.copy_(src)
    ev[1].record()
    torch.cuda.synchronize()
    return 2 * n * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12


print(f"copy of {gib} GiB buffers inside one allocation; distance = buffer size + d")
ds = [0] + [1 << k for k in range(8, 31)] + [3 << k for k in range(8, 29, 2)] + [(1 << 21) + (1 << 12), (1 << 24) + (1 << 16) + 256]
for d in ds:
    if n + d + n > slab.numel():
        continue
    print(f"d = {d:>11d} ({d / (1 << 20):10.4f} MiB): {rate(n + d):.3f} TB/s", flush=True)
# separate allocations, re-allocated a few times
for rep in range(6):
    a = torch.empty(n, dtype=torch.uint8, device="cuda").view(torch.float64)
    b = torch.empty(n, dtype=torch.uint8, device="cuda").view(torch.float64)
    a.zero_(); b.zero_()
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(12):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    print(f"separate allocations #{rep}: {2 * n * 12 / (e0.elapsed_time(e1) * 1e-3) / 1e12:.3f} TB/s  (a {a.data_ptr():#x} b {b.data_ptr():#x})", flush=True)
    keep = torch.empty(int((rep + 1) * 0.37 * (1 << 30)), dtype=torch.uint8, device="cuda")   # shift the next pair
    del a, b
    torch.cuda.empty_cache()
