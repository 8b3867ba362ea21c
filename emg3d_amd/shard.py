"""Frequency sharding across the GPUs of one node (reference counterpart:
``Simulation.compute`` fans (source, frequency) pairs out over a process pool,
emg3d/simulations.py:821-878).  Every (source, frequency) pair is an
independent linear system, so the path shards with NO data-path collective:
rank r solves the frequencies ``freqs[r::world]`` on GPU r; the only exchange
is ONE end-of-run gather of the fields (RCCL over xGMI when the backend is
"nccl"; gloo on CPU for the tests).
"""
import numpy as np


def my_frequencies(freqs, rank, world):
    """Round-robin assignment: rank r gets freqs[r], freqs[r+world], ..."""
    return [float(f) for f in list(freqs)[rank::world]]


def solve_frequencies(grid, model, src, freqs, device=0, strength=0, concurrent=1, **solver_opts):
    """Solve one source for several frequencies on ONE GPU from a shared, frequency-independent
    model: ``sigma*V`` and ``zeta`` are computed once (``models.sigma_volume``); per frequency only
    the scalar ``s*mu_0`` changes (``eta = s mu_0 sigma V`` and the source ``s mu_0 * vector`` are
    formed on the device).  Reference counterpart: the per-frequency jobs of
    ``Simulation.compute`` (emg3d/simulations.py:840-867) with ``gridding='same'``.

    ``concurrent`` > 1 runs that many solves at the same time on the GPU (the ``max_workers`` of the
    reference's process pool, simulations.py:862-867, as host threads: every solve has its own handle
    and HIP stream, the library calls release the GIL).  The coarse levels of a cycle leave most
    SIMDs idle, so a second and third frequency fill them: measured 184 -> 233 -> 256 Mcells/s
    aggregate for 1 -> 2 -> 3 concurrent 128^3 F-cycles on one MI355X (``tools/multi_solve.py``).
    Results do not depend on ``concurrent`` (each solve is deterministic on its own stream).
    Returns ``[(efield, info), ...]`` in the order of ``freqs``."""
    from emg3d_amd import fields, models, solver
    freqs = [float(f) for f in freqs]
    if not freqs:
        return []
    sv = models.sigma_volume(grid, model)
    # the real source vector does not depend on the frequency
    vector = fields.get_source_field(grid, src, freqs[0], strength=strength).vector

    def one(f):
        sfield = fields.SourceField(grid, freq=f)
        sfield.field[:] = sfield.smu0 * vector
        with solver.DeviceMG.from_sigma_volume(grid, *sv, smu0=sfield.smu0, device=device) as dev:
            return solver.solve(grid, None, sfield, handle=dev, return_info=True, **solver_opts)

    if int(concurrent) <= 1 or len(freqs) == 1:
        return [one(f) for f in freqs]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(int(concurrent), len(freqs))) as pool:
        return list(pool.map(one, freqs))


def gather_fields(local, group=None):
    """All-gather equally sized 1-D field arrays (complex128/float64).

    ``local``: list of NumPy arrays (this rank's fields, all the same length)
    or a single array.  Returns a list over ranks of lists of arrays.  Uses
    ``torch.distributed`` (nccl -> device tensors, gloo -> host tensors); with
    an un-initialised process group it degenerates to ``[local]``.
    """
    import torch
    import torch.distributed as dist
    single = isinstance(local, np.ndarray)
    arrs = [local] if single else list(local)
    if not (dist.is_available() and dist.is_initialized()):
        return [arrs]
    world = dist.get_world_size(group)
    cplx = np.iscomplexobj(arrs[0])
    n = arrs[0].size
    stack = np.stack([np.ascontiguousarray(a) for a in arrs])
    flat = torch.from_numpy(stack.view(np.float64).reshape(-1).copy())
    if dist.get_backend(group) == "nccl":
        flat = flat.cuda()
    # the number of fields may differ by one between ranks: gather counts first
    cnt = torch.tensor([len(arrs)], dtype=torch.int64, device=flat.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    cnts = [int(c.item()) for c in cnts]
    per = n * (2 if cplx else 1)
    pad = torch.zeros(max(cnts) * per, dtype=torch.float64, device=flat.device)
    pad[:flat.numel()] = flat
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for c, p in zip(cnts, parts):
        a = p[:c * per].cpu().numpy().reshape(c, per)
        a = a.view(np.complex128) if cplx else a
        out.append([a[i].copy() for i in range(c)])
    return out
