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


def solve_frequencies(grid, model, src, freqs, device=0, strength=0, concurrent=1, rec=None, return_field=True,
                      **solver_opts):
    """Solve one source for several frequencies on ONE GPU from a shared, frequency-independent
    model: conductivities, cell volumes and ``zeta`` are computed and uploaded once (``models.model_parts``); per
    frequency only the scalar ``s*mu_0`` changes (``eta = (s mu_0 V) sigma`` -- VolumeModel's rounding, so every result is
    bit for bit that of ``solver.solve`` at this frequency -- and the source are formed on the device).  Reference counterpart: the per-frequency jobs of
    ``Simulation.compute`` (emg3d/simulations.py:840-867) with ``gridding='same'``.

    ``concurrent`` > 1 runs that many solves at the same time on the GPU (the ``max_workers`` of the
    reference's process pool, simulations.py:862-867, as host threads: every solve has its own handle
    and HIP stream, the library calls release the GIL).  The coarse levels of a cycle leave most
    SIMDs idle, so a second and third frequency fill them: measured 184 -> 233 -> 256 Mcells/s
    aggregate for 1 -> 2 -> 3 concurrent 128^3 F-cycles on one MI355X (``tools/multi_solve.py``).
    Results do not depend on ``concurrent`` (each solve is deterministic on its own stream).

    ``rec = (x, y, z, azimuth, dip)``: also extract the receiver responses of every solution
    (``fields.get_receiver_response``, reference fields.py:733-817) -- straight from the field in HBM
    (``DeviceMG.get_receiver_response``); with ``return_field=False`` the field itself is never downloaded,
    so a rank hands 16 bytes per receiver to the end-of-run gather instead of 102 MB per frequency (128^3).

    Returns ``[(efield, info), ...]`` in the order of ``freqs``; with ``rec``: ``[(efield | None, info,
    responses), ...]``."""
    from emg3d_amd import fields, models, solver
    freqs = [float(f) for f in freqs]
    if not freqs:
        return []
    parts = models.model_parts(grid, model, raw=True)        # (with or without epsilon_r)

    def one(f, handles=None):
        # the source is built in HBM per frequency (DeviceMG.set_source: the dipole's edge distribution runs on the
        # device, scaled by this frequency's s mu_0): no nE-sized array is formed or uploaded on the host
        # (with a Krylov solver the host object carries the right-hand side; otherwise only the frequency)
        sfield = fields.SourceField(grid, freq=f) if solver_opts.get('sslsolver') else fields.FrequencySpec(f)
        # frequencies solved one after the other share ONE handle per dtype: only eta and what is derived from it
        # (coarse models, line factorisations) is recomputed (DeviceMG.set_smu0) -- the results are those of a fresh
        # handle bit for bit; hierarchy, buffers and launch graphs are not rebuilt (15-20 ms per frequency at 128^3)
        key = np.dtype(sfield.dtype).str
        dev = handles.get(key) if handles is not None and parts is not None else None
        if dev is None:
            if parts is not None:
                dev = solver.DeviceMG.from_model(grid, parts, sfield, device=device)
            else:
                dev = solver.DeviceMG(grid, models.VolumeModel(grid, model, sfield), sfield.dtype, device=device)
            if handles is not None and parts is not None:
                handles[key] = dev
        else:
            dev.set_smu0(sfield.smu0, sval=sfield.sval)
        try:
            # receivers only (multigrid path): the solution stays in HBM -- no nE-sized download that would be thrown away
            keep = not (rec is not None and not return_field and not solver_opts.get('sslsolver'))
            e, info = solver.solve(grid, None, sfield, handle=dev, return_info=True, source=(src, strength),
                                   download=keep, **solver_opts)
            if rec is None:
                return e, info
            if solver_opts.get('sslsolver'):       # the Krylov iterate lives in a workspace vector: use the host field
                resp = fields.get_receiver_response(grid, e, rec)
            else:
                resp = dev.get_receiver_response(rec)
            return (e if return_field else None), info, resp
        finally:
            if handles is None or parts is None:
                dev.close()

    if int(concurrent) <= 1 or len(freqs) == 1:
        handles = {}
        try:
            return [one(f, handles) for f in freqs]
        finally:
            for dev in handles.values():
                dev.close()
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(int(concurrent), len(freqs))) as pool:
        return list(pool.map(one, freqs))


def solve_survey(grid, model, sources, freqs, rec, device=0, strength=0, batch=8, return_fields=False, **solver_opts):
    """A survey's forward modelling on ONE GPU: every source of ``sources`` at every frequency of ``freqs`` (the
    (source, frequency) loop of ``Simulation.compute``, emg3d/simulations.py:821-878, with ``gridding='same'``) and
    the responses at the receivers ``rec = (x, y, z, azimuth, dip)``.  Frequencies are independent handles
    (across GPUs: give every rank ``my_frequencies(freqs, rank, world)`` and gather the responses with
    ``gather_fields``); the sources of one frequency share its operator and go through the SAME launches, ``batch``
    at a time (``solver.solve_sources``: bit for bit the results of one solve per source).  Nothing nE-sized
    crosses PCIe unless ``return_fields``: sources are built and receivers evaluated in HBM.

    Returns ``(responses, infos)`` -- ``responses[i_src, i_freq, i_rec]`` complex (or float for Laplace-domain
    frequencies), ``infos[i_src][i_freq]`` the ``info_dict`` of that solve -- and the fields ``[i_src][i_freq]`` if
    ``return_fields``."""
    from emg3d_amd import solver
    freqs = [float(f) for f in freqs]
    ns, nf = len(sources), len(freqs)
    resp = None
    infos = [[None] * nf for _ in range(ns)]
    efs = [[None] * nf for _ in range(ns)] if return_fields else None
    # one handle per (dtype, systems per launch), re-targeted from frequency to frequency (DeviceMG.set_smu0: eta, coarse
    # models and line factorisations are recomputed in HBM, hierarchy / buffers / launch graphs stay); bit for bit the
    # results of solver.solve_sources with a handle of its own
    from emg3d_amd import fields, models
    parts = models.model_parts(grid, model, raw=True)
    handles = {}
    try:
        for jf, f in enumerate(freqs):
            spec = fields.FrequencySpec(f)
            for i0 in range(0, ns, int(batch)):
                chunk = list(sources[i0:i0 + int(batch)])
                dev = None
                if parts is not None:
                    key = (np.dtype(spec.dtype).str, len(chunk))
                    dev = handles.get(key)
                    if dev is None:
                        dev = handles[key] = solver.DeviceMG.from_model(grid, parts, spec, device=device)
                        dev._smu0 = spec.smu0
                    elif dev._smu0 != spec.smu0:
                        dev.set_smu0(spec.smu0, sval=spec.sval)
                        dev._smu0 = spec.smu0
                e, info, r = solver.solve_sources(grid, model, chunk, f, strength=strength, rec=rec, device=device,
                                                  download=return_fields, handle=dev, **solver_opts)
                if resp is None:
                    resp = np.zeros((ns, nf, r.shape[1]), dtype=np.result_type(r.dtype, np.float64))
                if r.dtype.kind == 'c' and resp.dtype.kind != 'c':
                    resp = resp.astype(np.complex128)
                resp[i0:i0 + len(chunk), jf] = r
                for k in range(len(chunk)):
                    infos[i0 + k][jf] = info[k]
                    if return_fields:
                        efs[i0 + k][jf] = e[k]
    finally:
        for dev in handles.values():
            dev.close()
    if resp is None:            # this rank owns no frequency
        resp = np.zeros((ns, 0, int(max(np.size(c) for c in rec[:3]))))
    return (resp, infos, efs) if return_fields else (resp, infos)


def gather_survey(resp_local, freqs, group=None):
    """End-of-run exchange of a sharded survey ("optionally only receiver responses"): every rank holds
    ``resp_local[i_src, j, i_rec]`` for ITS frequencies ``my_frequencies(freqs, rank, world)`` (the output of
    ``solve_survey``; a rank without frequencies passes shape ``(n_src, 0, n_rec)``) and gets back the full
    ``(n_src, len(freqs), n_rec)`` array in the order of ``freqs``.  One header and one payload collective
    (``gather_fields``): 16 bytes per source, frequency and receiver cross the links."""
    import torch.distributed as dist
    freqs = [float(f) for f in freqs]
    resp_local = np.asarray(resp_local)
    if not (dist.is_available() and dist.is_initialized()):
        return resp_local
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n_mine = len(my_frequencies(freqs, rank, world))
    if resp_local.ndim != 3 or resp_local.shape[1] != n_mine:
        raise ValueError(f"gather_survey: rank {rank} owns {n_mine} frequencies, got an array of shape {resp_local.shape}.")
    ns, _, nr = resp_local.shape
    parts = gather_fields([np.ascontiguousarray(resp_local[:, j, :]).ravel() for j in range(n_mine)], group=group)
    cplx = any(np.iscomplexobj(a) for p in parts for a in p)
    out = np.zeros((ns, len(freqs), nr), dtype=np.complex128 if cplx else np.float64)
    for r in range(world):
        for j, a in enumerate(parts[r]):
            out[:, r + j * world, :] = a.reshape(ns, nr)          # my_frequencies(freqs, r, world) = freqs[r::world]
    return out


def gather_fields(local, group=None, device=None):
    """All-gather equally sized 1-D field arrays (complex128/float64).

    ``local``: list of NumPy arrays (this rank's fields, all the same length; may be EMPTY: a rank
    owns no frequency when there are fewer frequencies than ranks) or a single array.  Returns a list
    over ranks of lists of arrays.  Uses ``torch.distributed`` (nccl -> device tensors, gloo -> host
    tensors); with an un-initialised process group it degenerates to ``[local]``.

    The ranks first agree on (count, length, complex?) through a small header all_gather, so that a
    rank without fields takes part in the payload collective with a zero-filled buffer instead of
    raising before it (which would leave the other ranks blocked in the collective).

    ``device`` (nccl only): the GPU whose memory stages the collective -- this rank's own (``DeviceMG.device`` / LOCAL_RANK);
    default: the calling thread's current device (the library's stateless calls leave it unchanged).
    """
    import torch
    import torch.distributed as dist
    single = isinstance(local, np.ndarray)
    arrs = [local] if single else list(local)
    if not (dist.is_available() and dist.is_initialized()):
        return [arrs]
    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device)) if on_gpu else torch.device("cpu")
    have = len(arrs) > 0
    head = torch.tensor([len(arrs), arrs[0].size if have else 0,
                         int(np.iscomplexobj(arrs[0])) if have else 0], dtype=torch.int64, device=device)
    heads = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    heads = [[int(v) for v in h_.tolist()] for h_ in heads]
    cnts = [h_[0] for h_ in heads]
    owners = [h_ for h_ in heads if h_[0] > 0]
    if not owners:
        return [[] for _ in range(world)]
    n, cplx = owners[0][1], bool(owners[0][2])
    if any(h_[1] != n or bool(h_[2]) != cplx for h_ in owners):
        raise ValueError(f"gather_fields: ranks disagree on field length / dtype: {heads}")
    per = n * (2 if cplx else 1)
    pad = torch.zeros(max(cnts) * per, dtype=torch.float64, device=device)
    if have:
        stack = np.stack([np.ascontiguousarray(a) for a in arrs])
        flat = torch.from_numpy(stack.view(np.float64).reshape(-1).copy())
        pad[:flat.numel()] = flat.to(device)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for c, p in zip(cnts, parts):
        a = p[:c * per].cpu().numpy().reshape(c, per)
        a = a.view(np.complex128) if cplx else a
        out.append([a[i].copy() for i in range(c)])
    return out


class _DevArray:
    """A device buffer owned by a multigrid handle, seen through ``__cuda_array_interface__`` so that
    ``torch.as_tensor`` wraps it WITHOUT a copy (float64 view: complex128 = 2 doubles per entry)."""

    def __init__(self, ptr, n_doubles, owner):
        self.__cuda_array_interface__ = {"shape": (int(n_doubles),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner       # keeps the handle (and with it the memory) alive


def efield_tensor(dev):
    """The handle's level-0 electric field in HBM as a torch float64 tensor (zero copy; re/im
    interleaved for complex128 handles).  C ABI: ``emg3d_mg_efield_devptr`` / ``emg3d_mg_nE``.

    A SNAPSHOT: large levels keep the field in a parity-split working copy between cycles, and the call converts it back
    into the reference-layout buffer the tensor wraps (on the handle's stream; torch's CURRENT stream is made to wait for
    that stream before the tensor is returned, so torch work enqueued afterwards sees the converted field; consumers on
    other streams order themselves behind ``dev.stream_ptr``).  Valid until the next cycle / smoothing / solve call on the
    handle; call again afterwards instead of keeping the tensor, and do not write through it."""
    import torch
    from emg3d_amd._lib import HipLibraryError
    per = 2 if dev.dtype == np.complex128 else 1
    ptr = dev.efield_devptr         # (enqueues the conversion to the reference layout on the handle's stream)
    if not ptr:
        raise HipLibraryError("emg3d_mg_efield_devptr returned NULL: the handle is closed or a device call failed")
    t = torch.as_tensor(_DevArray(ptr, dev.nE * per, dev), device=torch.device("cuda", dev.device))
    # a torch consumer runs on torch's current stream: order it behind the handle's stream (no host synchronisation)
    torch.cuda.current_stream(t.device).wait_stream(torch.cuda.ExternalStream(dev.stream_ptr, device=t.device))
    return t


def gather_efield_device(dev, group=None):
    """Device-resident end-of-run gather: ONE ``all_gather_into_tensor`` of the handle's
    field straight out of its HBM buffer over RCCL/xGMI -- no host staging, no copy of the input.
    The collective is ordered behind the handle's own HIP stream (``emg3d_mg_stream``) by a stream wait, so no host
    synchronisation separates the last cycle from the gather.  Returns a (world, nE [*2]) float64
    device tensor (row r = rank r's field); ``.view(torch.complex128)`` rows for complex handles."""
    import torch
    import torch.distributed as dist
    src = efield_tensor(dev)
    if not (dist.is_available() and dist.is_initialized()):
        return src.clone().unsqueeze(0)
    world = dist.get_world_size(group)
    out = torch.empty((world, src.numel()), dtype=torch.float64, device=src.device)
    # The collective runs on torch's current stream, ordered behind the handle's stream by stream waits in both
    # directions (no host synchronisation).  It is NOT enqueued on the handle's stream itself: the process group's
    # watchdog thread keeps querying the events it records on the stream of a collective, and a later handle
    # whose stream the runtime gives the same address may be capturing a graph by then (hipErrorCapturedEvent).
    ext = torch.cuda.ExternalStream(dev.stream_ptr, device=src.device)
    cur = torch.cuda.current_stream(src.device)
    cur.wait_stream(ext)
    dist.all_gather_into_tensor(out.view(-1), src, group=group)    # flat output: the form every backend accepts
    ext.wait_stream(cur)        # later cycles of the handle do not overwrite the field under the collective
    return out
