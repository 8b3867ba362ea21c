#!/usr/bin/env python
"""Benchmark of the multigrid hot path on MI355X (contract: see task README).

  python bench.py --gpus N --steps K --warmup W [--workload 128F|256V|64F] [--ordering colour|lex]

A "step" is ONE multigrid cycle (one level-0 iteration of solver.multigrid:
pre-smoothing, residual, restriction, coarse-grid recursion, prolongation,
post-smoothing, end-of-cycle residual norm) on a problem that is resident in
HBM.  Metric: Mcells/s per cycle = fine-grid cells / time per cycle -- the
quantity the reference records as diff(info['runtime_at_cycle'])
(emg3d/solver.py:1588).  With N > 1 every rank runs the same grid for its own
source frequency (independent systems, no data-path collective; "weak"
scaling); the efields are gathered once at the end over RCCL (outside the timed
region, time reported).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment
spawns the N ranks itself (fresh child processes, before this process touches
the GPU); under `torch.distributed.run` the ranks come from the environment.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# N ranks on one node: every rank keeps its host-side BLAS / OpenMP pools at ONE thread.  A multi-threaded host BLAS call
# next to the GPU path stalls the process's GPU queues for 60-80 ms, and 8 ranks x 64-128 pool threads
# oversubscribe the host.  Must happen before NumPy is imported (spawn_ranks sets the same for its children).
_PIN = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "VECLIB_MAXIMUM_THREADS")
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    for _v in _PIN:
        os.environ.setdefault(_v, "1")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FREQS = [1.0, 0.25, 0.5, 0.75, 1.5, 2.0, 3.0, 4.0]   # rank r -> FREQS[r]

WORKLOADS = {
    # name: (n, ncore, npad, width, (fx, fy, fz), cycle)    SURVEY 8d C2 / C3
    "128F": (128, 64, 32, 50., (1.06, 1.06, 1.07), 'F'),
    "256V": (256, 128, 64, 25., (1.04, 1.04, 1.045), 'V'),
    "384V": (384, 192, 96, 16.667, (1.027, 1.027, 1.03), 'V'),     # capacity check: a 91 GB handle (not a BASELINE config)
    # beyond 4 GiB per field array (complex: ~445^3 and more): level 0 runs k_line_sweep_qc<..., BIG> (64-bit field offsets)
    "448V": (448, 224, 112, 14.286, (1.0228, 1.0228, 1.0255), 'V'),
    "512V": (512, 256, 128, 12.5, (1.02, 1.02, 1.0225), 'V'),      # 134 M cells, 403 M unknowns: the 288 GB of one MI355X
    # between the powers of two (launch shapes by rounds of waves, HISTORY R5.19): parity-test sizes
    "144V": (144, 72, 36, 44.4, (1.072, 1.072, 1.075), 'V'),       # 5184 lines per colour: two-sided kernel, 12 lines per pair of waves
    "200V": (200, 100, 50, 32., (1.052, 1.052, 1.055), 'V'),       # 10 000 lines per colour: quad kernel at 10 lines per wave
    "64F": (64, 32, 16, 100., (1.12, 1.12, 1.14), 'F'),
    "32F": (32, 16, 8, 200., (1.25, 1.25, 1.3), 'F'),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
SWEEP_BYTES_PER_CELL = 200.0   # e r+w 96, s 48, eta 48, zeta 8 (complex128, tri-axial)
RESID_BYTES_PER_CELL = 200.0
SWEEP_FLOP_PER_CELL = 1500.0   # minimal band LDL^T line sweep, complex128
FP64_PEAK_TFLOPS = 78.6        # MI355X vector FP64
SC_CYCLE, LR_CYCLE = [1, 2, 3], [4, 5, 6]
SWEEP_SAMPLES = 10             # isolated-sweep timings: event brackets per direction (min / median / max are reported)


def build_problem(em, name, freq):
    """Synthetic stretched-grid, tri-axial marine model."""
    n, ncore, npad, w, fac, cycle = WORKLOADS[name]
    h = [em.meshes.stretched_widths(ncore, npad, w, f) for f in fac]
    grid = em.TensorMesh(h, origin=[-hh.sum() / 2 for hh in h])
    X, Y, Z = np.meshgrid(grid.cell_centers_x, grid.cell_centers_y, grid.cell_centers_z, indexing='ij')
    rho = np.full(grid.vnC, 10.0)
    rho[Z > -1000.] = 1.0
    rho[Z > 0.] = 0.3
    rho[(abs(X) < 500) & (abs(Y) < 500) & (Z > -800) & (Z < -600)] = 100.0
    rho = rho.ravel(order='F')
    model = em.Model(grid, rho, 2 * rho, 3 * rho)
    sfield = em.get_source_field(grid, [0., 0., 0., 30., 10.], freq)
    return grid, model, sfield, cycle


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _cpu_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(em, workload, concurrent=True):
    """The reference's CPU path stand-in (numba is unavailable and the reference
    cannot travel): the oracle's C++ restatement, -O3 -ffast-math, ONE thread
    (the reference is single-threaded, emg3d/core.py:25), in the reference's
    lexicographic order, on a bounded sample of the SAME workload as the
    headline: two cycles of it (128^3: ~25 s).  Compiled on THIS machine with
    -march=native (`oracle.build_native`), the shipped -march=x86-64-v3 build if no compiler is here.

    `concurrent`: BASELINE configs[4]'s CPU counterpart -- min(8, cores) single-thread solves of the same grid at the
    frequencies of the eight-GPU shard side by side (the reference's process pool, emg3d/simulations.py:862-867
    `max_workers`; here host threads: the oracle's kernels are ctypes calls, which release the GIL)."""
    from oracle import oracle as orc
    grid, model, sfield, cycle = build_problem(em, workload, 1.0)
    native = orc.build_native() is not None
    fast = "native" if native else True
    ncyc = 2 if grid.nC >= 128 ** 3 else 7

    def one(freq):
        g, m, sf, cyc = build_problem(em, workload, freq)
        vm = em.VolumeModel(g, m, sf)
        t0 = time.perf_counter()
        _, info = orc.solve(orc.Mesh(g.h, g.origin), orc.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta), np.array(sf), cycle=cyc,
                            semicoarsening=True, linerelaxation=True, maxit=ncyc, tol=1e-30, order=0, fast=fast)
        return time.perf_counter() - t0, float(np.diff(info['runtime_at_cycle']).mean())

    wall, dt = one(1.0)
    out = {
        "value": float(grid.nC / dt / 1e6), "unit": "Mcells/s per cycle", "cores": 1,
        "kind": "port",
        "sample": f"{grid.vnC[0]}^3 stretched tri-axial (the headline workload), {ncyc} {cycle}-cycles sc+lr, "
                  f"lexicographic order, C++ -O3 -ffast-math single thread ({wall:.1f} s)",
        "s_per_cycle": dt,
        "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(), "cores_available": _cpu_cores(),
        "march": "native (compiled on this machine)" if native else "x86-64-v3 (the shipped build: no compiler on this machine)",
    }
    nw = min(8, _cpu_cores())
    if concurrent and nw > 1:
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nw) as pool:
            res = list(pool.map(one, FREQS[:nw]))
        wall_c = time.perf_counter() - t0
        per = [float(grid.nC / d / 1e6) for _, d in res]
        out["concurrent"] = {
            "workers": nw, "cores": nw, "value": float(sum(per)), "unit": "Mcells/s per cycle, aggregate",
            "per_worker": per, "wall_s": wall_c, "freqs_Hz": FREQS[:nw],
            "sample": f"BASELINE configs[4] on the host: {nw} single-thread solves side by side (one frequency each, the "
                      f"frequencies of the GPU shard), {ncyc} {cycle}-cycles each; the reference's max_workers "
                      f"(emg3d/simulations.py:862-867)"}
    return out


def cycle_alg_bytes(vnC, cycle, nu=(0, 2, 1, 2), cycmax=None, executed=False):
    """Algorithmic HBM bytes of ONE multigrid cycle (SURVEY 8d per-kernel figures x the visits of the cycle, mean over
    the three (sc_dir, lr_dir) states of the rotation): per visit of a level with c cells
    (nu_pre + nu_post) sweeps x line directions x 200 B c + residual 200 B c + restriction 54 B c + prolongation 102 B c
    (coarsest level: nu_coarse sweeps only), plus the end-of-cycle residual norm on level 0 (200 B c).  Level shapes and
    visit counts follow solver.multigrid (emg3d/solver.py:471-586, 1467-1572).

    executed=True: the colour passes the device really launches -- a line-smoothing call of nu sweeps skips the colour
    repeated at each turn-around (bit-identical: a line update is a projection), 3 nu + 1 passes instead of the
    reference's 4 nu (7 of 8 at nu = 2)."""
    cycmax = cycmax or (1 if cycle == 'V' else 2)

    def cur_sc(sc_dir, n):          # solver._current_sc_dir
        xs = n[0] % 2 != 0 or n[0] < 3 or sc_dir == 1
        ys = n[1] % 2 != 0 or n[1] < 3 or sc_dir == 2
        zs = n[2] % 2 != 0 or n[2] < 3 or sc_dir == 3
        if xs:
            return 6 if ys else (5 if zs else 1)
        if ys:
            return 4 if zs else 2
        return 3 if zs else 0

    def n_line_dirs(lr, n):         # solver._current_lr_dir -> number of line directions (0: point smoother = 1 sweep)
        if n[0] == 2:
            lr = {1: 0, 5: 3, 6: 2, 7: 4}.get(lr, lr)
        if n[1] == 2:
            lr = {2: 0, 4: 3, 6: 1, 7: 5}.get(lr, lr)
        if n[2] == 2:
            lr = {3: 0, 4: 2, 5: 1, 7: 6}.get(lr, lr)
        return {0: 1, 1: 1, 2: 1, 3: 1, 4: 2, 5: 2, 6: 2, 7: 3}[lr]

    def lr_is_point(lr, n):         # the point smoother serves (eight colours, the same order in every sweep: nothing skipped)
        if n[0] == 2:
            lr = {1: 0, 5: 3, 6: 2, 7: 4}.get(lr, lr)
        if n[1] == 2:
            lr = {2: 0, 4: 3, 6: 1, 7: 5}.get(lr, lr)
        if n[2] == 2:
            lr = {3: 0, 4: 2, 5: 1, 7: 6}.get(lr, lr)
        return lr == 0

    cl = []
    for n in vnC:
        c = 0
        while n % 2 == 0 and n > 2:
            c += 1
            n //= 2
        cl.append(c)
    clevel = [max(cl), max(cl[1], cl[2]), max(cl[0], cl[2]), max(cl[0], cl[1])]
    total = 0.0
    for g, lr in zip(SC_CYCLE, LR_CYCLE):
        shapes = [list(vnC)]
        for _ in range(clevel[g]):
            n = shapes[-1]
            sc = cur_sc(g, n)
            co = [sc not in (1, 5, 6), sc not in (2, 4, 6), sc not in (3, 4, 5)]
            shapes.append([n[a] // 2 if co[a] else n[a] for a in range(3)])
        visits = [0] * len(shapes)

        def rec(level, new_cycmax):
            cm = 1 if level == clevel[g] else (cycmax if (new_cycmax == 0 or cycle != 'F') else new_cycmax)
            for cyc in range(cm):
                visits[level] += 1
                if level < clevel[g]:
                    rec(level + 1, cm - cyc)
        visits[0] = 1
        if clevel[g] > 0:
            rec(1, 1 if clevel[g] == 0 else cycmax)
        b = 0.0
        for lev, (n, v) in enumerate(zip(shapes, visits)):
            c = n[0] * n[1] * n[2]
            nd = n_line_dirs(lr, n)
            point = lr_is_point(lr, n)

            def sw(k):          # sweeps' worth of bytes of one smoothing call of k sweeps
                return k if (not executed or point or k == 0) else (3 * k + 1) / 4.0
            if lev == clevel[g]:
                b += v * sw(nu[2]) * nd * SWEEP_BYTES_PER_CELL * c
            else:
                b += v * ((sw(nu[1]) + sw(nu[3])) * nd * SWEEP_BYTES_PER_CELL + RESID_BYTES_PER_CELL + 54.0 + 102.0) * c
        b += RESID_BYTES_PER_CELL * vnC[0] * vnC[1] * vnC[2]
        total += b / 3
    return total


def _build_info():
    try:
        with open(os.path.join(ROOT, "emg3d_amd", "build_info.json")) as fh:
            return json.load(fh)
    except Exception:
        return {}


def _code_id():
    """Which code runs: HEAD where the checkout is a repository (not on the GPU boxes, which get a snapshot without .git), and
    what __graft_entry__.build() recorded beside the library when it compiled it: the last commit that touched the sources
    and whether they were dirty then."""
    head = None
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:
        pass
    bi = _build_info()
    return {"head": head, "sources_commit": bi.get("sources_commit"), "sources_dirty": bi.get("dirty"), "built": bi.get("built")}


def _rocprof_average_ms(kname, which):
    """Average duration of `kname` in the newest committed `rocprofv3 --kernel-trace --stats` summary
    profiles/r*_<which>_kernel_stats.csv (`bench`: the bench command itself, dipole source; `sweep_<workload>_dense`: the
    isolated level-0 sweeps with a dense right-hand side), to put beside the HIP-event time of this run."""
    import csv
    import glob
    # isolated-sweep profiles: the average over the TIMED launches of that run (profiles/collect.sh: the last 360 dispatches of the
    # kernel; the set-up in front of them launches it on the placement search's candidate blocks) when the collection has it
    timed = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{which}_timed_launches.json")))
    stats = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{which}_kernel_stats.csv")))
    if timed and (not stats or os.path.basename(timed[-1])[:3] >= os.path.basename(stats[-1])[:3]):
        try:
            with open(timed[-1]) as fh:
                t = json.load(fh)
            if t.get("kernel", "").startswith(kname.rstrip(">")):
                return {"file": os.path.relpath(timed[-1], ROOT), "average_ms": t["average_ms"], "calls": t["timed_launches"],
                        "launches_of_the_kernel_in_the_run": t.get("launches_of_the_kernel_in_the_run"),
                        # the profiled process's own HIP-event time of those launches: the timers agree when this ratio is 1
                        "hip_event_ms_same_process": t.get("hip_event_ms_same_process"),
                        "profiler_over_hip_events": t.get("profiler_over_hip_events")}
        except (OSError, ValueError, KeyError):
            pass
    files = stats
    if not files:
        return None
    want = kname.replace(",", ", ").replace("  ", " ").rstrip(">")
    with open(files[-1]) as fh:
        for r in csv.DictReader(fh):
            if want in r["Name"]:
                return {"file": os.path.relpath(files[-1], ROOT), "average_ms": float(r["AverageNs"]) * 1e-6, "calls": int(r["Calls"])}
    return None


FORMULATION_BYTES_PER_BLOCK = {   # what an exact line solve with a cached factor moves per block (= cell x colour launch)
    # factor forward + backward, parked z (write + read), source, result, neighbour values with perfect sharing; zeta is
    # formed from the width vectors on level 0 of models without mu_r (the bench workloads) and not counted
    "k_line_sweep_qc": 176 + 176 + 160 + 80 + 80 + 112,       # compact factor (11 numbers)
    "k_line_sweep_thm": 240 + 224 + 160 + 80 + 80 + 112,      # full mirrored factor
}


def roofline_of(dev, grid, workload, sfield=None, dense_only=False):
    """Dominant kernel = the line-smoother substitution sweep, isolated on the level-0 grid and timed
    with HIP events on the handle's stream.  One sweep = 4 launches (one per colour); algorithmic
    bytes per launch = 200 B/cell * cells / 4.

    `frac` is priced on the launch with a DENSE right-hand side -- what every Krylov vector and every coarse level is, and
    what the 200 B/cell of the algorithmic figure contain (48 B of them the source).  The workload's own source is a
    dipole: all but a handful of lines are source-free, and the level-0 kernels skip the source loads of such lines
    (bit-identical results) -- that launch is reported beside it (`launch_ms_sparse_source`,
    `frac_sparse_source`), credited with the same algorithmic bytes although it moves fewer."""
    launches = 4
    samples, per = SWEEP_SAMPLES, 2        # SWEEP_SAMPLES event brackets of `per` sweeps (4 launches each) per direction

    def sample():
        """(median over the samples per direction, per-launch statistics over the samples: every sample = mean over the three
        directions' launches, as `rocprofv3 --kernel-trace --stats` averages them in `bench.py --mode sweep`)"""
        t = {d: [dev.time_sweep(d, per) for _ in range(samples)] for d in (1, 2, 3)}
        each = [sum(t[d][k] for d in (1, 2, 3)) / (3 * launches) for k in range(samples)]
        med = {d: float(np.median(t[d])) for d in (1, 2, 3)}
        return med, {"min": float(min(each)), "median": float(np.median(each)), "max": float(max(each)), "mean": float(np.mean(each)),
                     "samples": samples, "sweeps_per_sample": per}
    ms = st_sparse = None
    if not dense_only:
        ms, st_sparse = sample()
    dense_ms_d = st_dense = None
    if sfield is not None:
        rng = np.random.default_rng(5)
        dense = np.array(sfield).copy()
        dense[:] = (rng.standard_normal(dense.size) + (1j * rng.standard_normal(dense.size) if np.iscomplexobj(dense) else 0)) * 1e-9
        dev.set_sfield(dense)
        dense_ms_d, st_dense = sample()
        dev.set_sfield(sfield)
    kname = dev.last_sweep_kernel()            # the instantiation the launch selection picked
    # duration of ONE launch of the sweep kernel: median over the samples of the mean over the level-0 sweeps of all three
    # directions; the conversions to / from the working copies are separate kernels outside the events.
    sparse_ms = st_sparse["median"] if st_sparse else None
    dense_ms = st_dense["median"] if st_dense else None
    launch_ms = dense_ms if dense_ms is not None else sparse_ms
    alg = SWEEP_BYTES_PER_CELL * grid.nC / launches
    ach = alg / (launch_ms * 1e-3) / 1e9
    # HBM bytes per launch: rocprofv3 --pmc passes of `bench.py --mode sweep`, collected by profiles/collect.sh in the same
    # gpurun call as the committed bench lines and summarised into profiles/traffic.json (which kernel, which sources:
    # traffic_source); `traffic_stale` when the library that runs now was built from other sources
    traffic, traffic_sparse, traffic_source, stale = None, None, None, None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        with open(tj) as fh:
            tjs = json.load(fh)
        ent = tjs.get(workload, {})
        # (rocprof prints every template argument, the library's own name the leading ones: "k<c128,3,8,0>" vs "k<c128,3,8>")
        if ent.get("kernel") is None or ent["kernel"].startswith(kname.rstrip(">")):
            traffic = ent.get("hbm_bytes_per_launch_dense_source")
            traffic_sparse = ent.get("hbm_bytes_per_launch")
            if dense_ms is None:
                traffic = traffic_sparse
        src = tjs.get("source", {})
        bi = _build_info()
        stale = bool(src.get("sources_commit") != bi.get("sources_commit") or src.get("sources_dirty") or bi.get("dirty"))
        traffic_source = {"file": "profiles/traffic.json", "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes "
                          "of `bench.py --mode sweep [--source dense]`; 2 x FETCH_SIZE + WRITE_SIZE (KiB), mean over the launches",
                          "kernel": ent.get("kernel"), "measured_in_this_run": False, **src}
    # FP64 co-limit: minimal band-LDL^T line sweep = 1.5 kflop per cell
    flops = SWEEP_FLOP_PER_CELL * grid.nC / launches / (launch_ms * 1e-3) / 1e12
    out = {
        "kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": stale, "traffic_source": traffic_source,
        "launch_ms": launch_ms, "launch_ms_stats": st_dense if st_dense is not None else st_sparse,
        "launch_ms_stats_sparse_source": st_sparse if st_dense is not None else None, "launches_per_sweep": launches,
        "source": "dense right-hand side (every line carries a source: Krylov vectors, coarse levels)" if dense_ms is not None
                  else "dipole of the workload (sparse: source-free lines skip the source loads)",
        "launch_ms_sparse_source": sparse_ms if dense_ms is not None else None,
        "frac_sparse_source": (alg / (sparse_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (sparse_ms and dense_ms is not None) else None,
        "traffic_sparse_source": traffic_sparse if dense_ms is not None else None,
        "kernel_time_source": "hipEvent (events on the handle's stream around isolated level-0 sweeps, this run)",
        "rocprof_average": _rocprof_average_ms(kname, f"sweep_{workload}_dense" if dense_ms is not None else "bench"),
        "rocprof_average_sparse_source": _rocprof_average_ms(kname, "bench") if dense_ms is not None else None,
        "sweep_ms": ({"x": dense_ms_d[1], "y": dense_ms_d[2], "z": dense_ms_d[3]} if dense_ms_d else {"x": ms[1], "y": ms[2], "z": ms[3]}),
        "alg_bytes_per_launch": alg,
        "fp64": {"alg_flop_per_cell": SWEEP_FLOP_PER_CELL, "achieved": flops, "peak": FP64_PEAK_TFLOPS,
                 "unit": "TFLOP/s", "frac": flops / FP64_PEAK_TFLOPS},
    }
    fb = next((v for k, v in FORMULATION_BYTES_PER_BLOCK.items() if kname.startswith(k)), None)
    if fb:
        out["formulation_bytes_per_block"] = fb
    # self-check against the committed rocprofv3 summary: this run's HIP-event launch time over the profile's average.  Outside
    # [0.97, 1.03] the two were not the same state of the box: at 256^3 the launch moves by up to 10 % with where the blocks it
    # writes lie in physical memory (HISTORY R5.18; `placement` = what the handle did about it), at 128^3 boxes differ by 1-2 %.
    ra = out["rocprof_average"]
    if ra and ra.get("average_ms"):
        out["vs_profile"] = launch_ms / ra["average_ms"]
        if not 0.97 <= out["vs_profile"] <= 1.03:
            out["placement_mode"] = ("faster than the committed profile's process" if out["vs_profile"] < 1 else
                                     "slower than the committed profile's process") + " (physical placement / box: HISTORY R5.18, R6.1)"
    pl = dev.placement()
    if pl:
        # candidates timed per working copy ('x': the copy the x-line sweeps write, 'yz': the y- / z-line sweeps'), ms per LAUNCH (a
        # quarter of the timed sweep, the source of that moment) of the first candidate and of the kept one
        timed = [v for v in pl.values() if v.get("tries")]
        out["placement"] = {"tries": max([v["tries"] for v in timed] or [0]),
                            "first_ms": (sum(v["first_ms"] for v in timed) / len(timed) / launches) if timed else None,
                            "kept_ms": (sum(v["kept_ms"] for v in timed) / len(timed) / launches) if timed else None,
                            "reused": any(v.get("reused") for v in pl.values()),
                            "unit": "ms per launch, mean over the working copies", "per_working_copy": pl}
    return out


def formulation_floor(r, ncells, copy_gbs):
    """The ceiling of THIS formulation (an exact line solve with a cached factor, one colour per launch): its bytes per block
    (factor in both substitution passes, parked z, source, result, neighbours, zeta) moved at the copy rate
    plain streaming kernels reach on this box -- as a launch time and as the fraction of the 8 TB/s roofline the 200
    algorithmic B/cell would then show.  `frac` cannot exceed `formulation_floor_frac` without another formulation."""
    fb = r.get("formulation_bytes_per_block")
    if not fb:
        return
    floor_bytes = fb * ncells / r["launches_per_sweep"]
    floor_ms = floor_bytes / (copy_gbs * 1e9) * 1e3
    r["formulation_floor_ms_at_copy_rate"] = floor_ms
    r["formulation_floor_frac"] = r["alg_bytes_per_launch"] / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    r["launch_ms_vs_formulation_floor"] = r["launch_ms"] / floor_ms


def stream_rates():
    """What plain streaming kernels reach on THIS box (torch elementwise kernels on 1 GiB buffers): context for
    `roofline.peak` (the 8 TB/s of the data sheet) -- the sweeps' counted traffic moves at about the copy rate."""
    import torch
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            fn()
        t1.record(); torch.cuda.synchronize()
        return t0.elapsed_time(t1) / reps * 1e-3
    gb = n * 8 / 1e9
    out = {"read_GBs": gb / timed(lambda: a.sum()), "write_GBs": gb / timed(lambda: b.fill_(1.0)),
           "copy_GBs": 2 * gb / timed(lambda: b.copy_(a)), "note": "torch sum / fill_ / copy_ on 1 GiB float64 buffers"}
    del a, b
    torch.cuda.empty_cache()
    return out


def time_to_tol(em, workload, tol=1e-6):
    """Whole `solve()` to `tol` in both orderings (second solve of the process: device blocks come from
    the pool): cycles and seconds, so that the colour ordering's extra cycles are priced in."""
    grid, model, sfield, cycle = build_problem(em, workload, 1.0)
    out = {}
    for ordering in ("colour", "lex"):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            _, info = em.solve(grid, model, sfield, cycle=cycle, semicoarsening=True, linerelaxation=True,
                               tol=tol, verb=0, return_info=True, ordering=ordering)
            t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
        rt = np.asarray(info['runtime_at_cycle'])
        out[ordering] = {"cycles_to_tol": int(info['it_mg']), "s_to_tol": best,
                         "ms_per_cycle": 1e3 * float(np.diff(rt).mean()) if rt.size > 1 else None,
                         "rel_error": float(info['rel_error']), "exit": int(info['exit'])}
    # BASELINE configs[3]: the same problem with the multigrid cycle as preconditioner of BiCGSTAB (device resident)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        _, info = em.solve(grid, model, sfield, cycle=cycle, semicoarsening=True, linerelaxation=True, sslsolver='bicgstab',
                           tol=tol, verb=0, return_info=True)
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    out["bicgstab_colour"] = {"solver_steps": int(info['it_ssl']), "mg_cycles": int(info['it_mg']), "s_to_tol": best,
                              "rel_error": float(info['rel_error']), "exit": int(info['exit'])}
    out["tol"] = tol
    return out


def parity_16(em):
    """The north star's "same residual norm as the reference within 1e-10 relative", measured in this run: the committed 16^3
    solves (tests/golden/solves_16_colour.npz: the reference's own solver.solve with its smoothing calls replaced by the replay of
    the device's colour schedule in reference arithmetic; solves_16.npz: the reference as it is, lexicographic) against the HIP
    path -- the worst per-cycle deviation of the residual norm over ALL cycles (relative to that cycle's own norm, and
    relative to the source norm) and over the cycles inside the window the tests hold to 1e-10 (residual above 1e-5 of the source
    norm, tests/conftest.py::assert_norms_close; later cycles carry the cancellation error of s - A e, which is relative to ||s||)."""
    gold = os.path.join(ROOT, "tests", "golden")
    try:
        g = np.load(os.path.join(gold, "solves_16.npz"))
        c = np.load(os.path.join(gold, "solves_16_colour.npz"))
    except OSError:
        return None
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    sfield = em.get_source_field(grid, g['src'], float(g['freq']))
    out = {"strict_window": "cycles whose residual norm is above 1e-5 x the source norm (tests/conftest.py: 1e-10 there, "
                            "2e-9 + 1e-14 ||s|| later)", "cases": {}}
    worst = {"all": 0.0, "window": 0.0, "vs_source": 0.0, "field": 0.0}
    for ordering, fix in (("colour", c), ("lex", g)):
        for name, kw in (("F_sclr", dict(cycle='F', semicoarsening=True, linerelaxation=True)),
                         ("V_sclr", dict(cycle='V', semicoarsening=True, linerelaxation=True))):
            e, info = em.solve(grid, model, sfield, return_info=True, ordering=ordering, verb=0, **kw)
            got, ref = np.asarray(info['error_at_cycle'], float), np.asarray(fix[f'{name}_error_at_cycle'], float)
            if got.shape != ref.shape:
                out["cases"][f"{ordering}_{name}"] = {"error": f"{got.size} cycles, fixture {ref.size}"}
                worst = {k: float("inf") for k in worst}
                continue
            dev_ = np.abs(got - ref) / np.abs(ref)
            win = np.abs(ref) > 1e-5 * abs(ref[0])
            fe = float(np.abs(np.asarray(e) - fix[f'{name}_efield']).max() / np.abs(fix[f'{name}_efield']).max())
            rec = {"cycles": int(info['it_mg']), "cycles_reference": int(fix[f'{name}_it'][0]),
                   "max_norm_dev_all_cycles": float(dev_.max()), "max_norm_dev_strict_window": float(dev_[win].max()),
                   "max_norm_dev_vs_source_norm": float((np.abs(got - ref) / abs(ref[0])).max()), "field_rel_dev": fe}
            out["cases"][f"{ordering}_{name}"] = rec
            worst["all"] = max(worst["all"], rec["max_norm_dev_all_cycles"])
            worst["window"] = max(worst["window"], rec["max_norm_dev_strict_window"])
            worst["vs_source"] = max(worst["vs_source"], rec["max_norm_dev_vs_source_norm"])
            worst["field"] = max(worst["field"], fe)
    out.update({"max_norm_dev_all_cycles": worst["all"], "max_norm_dev_strict_window": worst["window"],
                "max_norm_dev_vs_source_norm": worst["vs_source"], "max_field_rel_dev": worst["field"],
                "fixtures": "tests/golden/solves_16_colour.npz (timed colour ordering, reference arithmetic replay), "
                            "tests/golden/solves_16.npz (reference, lexicographic); 16^3 stretched tri-axial, F- and V-cycles sc+lr"})
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (this
    process has not initialised the GPU), relay rank 0's JSON line, fail if any rank fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        for v in _PIN:              # one host thread per rank for BLAS / OpenMP pools (set before the child imports NumPy)
            env.setdefault(v, "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = procs[0].communicate()[0]
    codes = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit(f"bench.py: ranks failed (rank, exit code): {bad}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="128F", choices=list(WORKLOADS))
    ap.add_argument("--ordering", default="colour", choices=["colour", "lex"])
    ap.add_argument("--mode", default="cycle", choices=["cycle", "sweep"],
                    help="'sweep': only the isolated kernel timings (for rocprofv3 agreement)")
    ap.add_argument("--source", default="dipole", choices=["dipole", "dense"],
                    help="--mode sweep: time the level-0 sweeps with the workload's dipole source or with a dense right-hand side")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-256", action="store_true", help="skip the config_256V object (256^3 V-cycle roofline config)")
    ap.add_argument("--no-tol", action="store_true", help="skip the time-to-tolerance solves of both orderings")
    ap.add_argument("--no-dense", action="store_true",
                    help="skip `roofline.launch_ms_dense_source` (the level-0 sweeps again with a dense right-hand side): keeps the "
                         "per-kernel averages of a `rocprofv3 --stats` run of this command those of the workload's own launches")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the isolated level-0 sweeps and the residual timing (profiles of the cycles alone: `rocprofv3 --stats` "
                         "of such a run counts the launches of the timed cycles and of the set-up only)")
    ap.add_argument("--multi", type=int, default=0,
                    help="N=1 only: also report the aggregate rate of this many concurrent solves (other "
                         "frequencies, own handles and streams) on the one GPU; 0 = skip (default: the kernels of "
                         "concurrent solves slow each other down, which would blur the per-kernel averages that "
                         "`rocprofv3 --stats` of this command must reproduce)")
    ap.add_argument("--batch", default="4",
                    help="N=1 only: also report the aggregate rate of this many SOURCES carried through the same launches "
                         "(DeviceMG.set_batch: shared model and line factorisations), comma-separated list; 0 = skip")
    ap.add_argument("--batch-tune", action="store_true",
                    help="batched_sources: also measure with EMG3D_BATCH_TUNE=1 (coarse-level kernel choice by lines x systems; "
                         "not in the default line: it runs the level-0 kernel template on other levels too, which would blur "
                         "the per-kernel averages that `rocprofv3 --stats` of this command must reproduce)")
    ap.add_argument("--echo-env", action="store_true",
                    help="harness self-test (no GPU): every rank reports its rank environment and exits")
    ap.add_argument("--fail-rank", type=int, default=-1, help="harness self-test: this rank exits with code 3")
    ap.add_argument("--freq-offset", type=int, default=0,
                    help="rank r solves FREQS[(r + offset) % 8] (tests: the single-rank run of another rank's frequency)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.echo_env:
        if rank == args.fail_rank:
            raise SystemExit(3)
        if rank == 0:
            print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world,
                              "master": os.environ.get("MASTER_ADDR"), "port": os.environ.get("MASTER_PORT"),
                              "threads": {v: os.environ.get(v) for v in _PIN}}))
        return

    import torch   # first: its HIP runtime is the one the process uses
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # harness self-test on a box with fewer GPUs than ranks: EMG3D_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and uses the
    # gloo backend (RCCL refuses two ranks on one device); everything else is the code path of the real run
    share = os.environ.get("EMG3D_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("EMG3D_FORCE_DIST") == "1"   # 1-rank RCCL self-test
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import emg3d_amd as em
    from emg3d_amd.solver import DeviceMG, MGParameters

    freq = FREQS[(rank + args.freq_offset) % len(FREQS)]
    grid, model, sfield, cycle = build_problem(em, args.workload, freq)
    vm = em.VolumeModel(grid, model, sfield)
    var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True,
                       vnC=grid.vnC, ordering=args.ordering)
    dev = DeviceMG(grid, vm, sfield.dtype, device=local_rank)
    dev.set_params(var)
    dev.set_sfield(sfield)
    dev.set_efield(None)
    l2_refe = dev.sfield_norm()     # on the device; a multi-threaded host BLAS norm stalls the GPU queues later

    def sync():
        dev._lib.emg3d_mg_sync(dev._h)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    out = {}
    if args.mode == "cycle":
        t_setup0 = time.perf_counter()
        # loop-invariant set-up (hierarchies, transfer weights, factor caches, captured launch
        # sequences) for the three (sc_dir, lr_dir) states of the rotation: outside the timed region
        # whatever --warmup is, reported as setup_plus_warmup_s
        for sc, lr in zip(SC_CYCLE, LR_CYCLE):
            dev.prepare(sc, lr)
        if args.warmup > 0:
            norms_w = dev.cycles(args.warmup, SC_CYCLE, LR_CYCLE)
        sync()
        t_setup = time.perf_counter() - t_setup0
        # continue the rotation where the warm-up stopped
        rot = args.warmup % 3
        sync()
        t0 = time.perf_counter()
        norms = dev.cycles(args.steps, SC_CYCLE[rot:] + SC_CYCLE[:rot], LR_CYCLE[rot:] + LR_CYCLE[:rot])
        sync()
        t = time.perf_counter() - t0
        tt = torch.tensor([t], device="cuda", dtype=torch.float64)
        per_rank = [t]
        if use_dist:
            allt = torch.zeros(world, device="cuda", dtype=torch.float64)
            dist.all_gather_into_tensor(allt, tt)
            per_rank = [float(x) for x in allt.tolist()]
        t_max = max(per_rank)
        ms_per_step = 1e3 * t_max / args.steps
        value = world * grid.nC * args.steps / t_max / 1e6
        hist = np.r_[norms_w if args.warmup else [], norms] / l2_refe
        below = np.nonzero(hist < 1e-6)[0]
        per_rank_hist = [[float(x) for x in hist]]
        if use_dist:                # every rank's residual history (its own frequency) in rank 0's line
            allh = torch.zeros(world * hist.size, device="cuda", dtype=torch.float64)
            dist.all_gather_into_tensor(allh, torch.tensor(hist, device="cuda", dtype=torch.float64))
            per_rank_hist = [[float(x) for x in row] for row in allh.reshape(world, hist.size).tolist()]

        # which devices the ranks really ran on, and how many ranks the process group saw (an N > 1 record must show that RCCL
        # had N ranks on N different GPUs)
        me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
              "name": torch.cuda.get_device_name(torch.cuda.current_device()),
              "uuid": str(getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "uuid", "")),
              "host": socket.gethostname()}
        per_rank_dev = [me]
        if use_dist:
            try:
                per_rank_dev = [None] * world
                dist.all_gather_object(per_rank_dev, me)
            except Exception as exc:        # (a record, not part of the measurement: never fail the run for it)
                per_rank_dev = [me, {"error": f"all_gather_object failed: {exc!r}"}]

        def first_below(h, tol=1e-6):
            idx = [i for i, x in enumerate(h) if x < tol]
            return idx[0] + 1 if idx else None
        out.update({
            "metric": "Mcells/s per multigrid cycle", "value": value, "unit": "Mcells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "c128",
            "data": "synthetic",
            "config": {"workload": f"{grid.vnC[0]}x{grid.vnC[1]}x{grid.vnC[2]} stretched grid, tri-axial "
                                   f"anisotropy, {cycle}-cycle, semicoarsening+linerelaxation, "
                                   f"nu=0/2/1/2, one frequency per GPU (rank0: {freq} Hz)",
                       "ordering": args.ordering, "cells": int(grid.nC)},
            "rel_error_after": [float(x) for x in hist],
            "cycles_to_1e-6": int(below[0]) + 1 if below.size else None,
            "per_rank_ms_per_step": [1e3 * x / args.steps for x in per_rank],
            "per_rank_freq_Hz": [FREQS[(r + args.freq_offset) % len(FREQS)] for r in range(world)],
            "per_rank_cycles_to_tol": [first_below(h) for h in per_rank_hist],
            "per_rank_rel_error_after": per_rank_hist,
            "host_threads_per_rank": os.environ.get("OMP_NUM_THREADS"),
            "rccl_world": (dist.get_world_size() if use_dist else None),
            "dist_backend": (dist.get_backend() if use_dist else None),
            "device_count": torch.cuda.device_count(),
            "per_rank_device": per_rank_dev,
            "setup_plus_warmup_s": t_setup,
            "device_GB": dev.device_bytes / 1e9,
        })
        cab = cycle_alg_bytes(grid.vnC, cycle)
        cabx = cycle_alg_bytes(grid.vnC, cycle, executed=(args.ordering == "colour"))
        out["cycle_algorithmic"] = {"bytes_per_cycle": cab, "GBs": cab / (ms_per_step * 1e-3) / 1e9,
                                    "frac": cab / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "bytes_per_cycle_executed": cabx,
                                    "frac_executed": cabx / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "whole cycle: algorithmic bytes of every sweep, residual and transfer of the "
                                            "cycle (bench.cycle_alg_bytes) / ms_per_step / 8 TB/s.  `frac` counts the "
                                            "reference's 4 colour passes per sweep (useful work), `frac_executed` the 3 nu + 1 "
                                            "passes per smoothing call that are launched (the repeated turn-around colour is "
                                            "skipped, bit-identically)"}
        out["code"] = dict(_code_id(), library=os.path.basename(em._lib.LIB_PATH))
        # final gather of the fields over RCCL/xGMI (outside the timed region): device resident, straight out
        # of the handle's HBM buffer, ordered behind the handle's stream by a stream wait
        if use_dist:
            from emg3d_amd import shard
            torch.cuda.synchronize(); dist.barrier()
            tg = time.perf_counter()
            allf = shard.gather_efield_device(dev)      # ONE all_gather_into_tensor over RCCL/xGMI
            torch.cuda.synchronize()
            out["gather_ms"] = 1e3 * (time.perf_counter() - tg)
            out["gather_bytes_per_rank"] = int(allf.shape[1] * 8)
            # the job including its one collective: N ranks' cells x K cycles over (slowest rank's cycles + the gather)
            out["value_incl_gather"] = world * grid.nC * args.steps / (t_max + out["gather_ms"] * 1e-3) / 1e6
            assert allf.shape[0] == world
            mine = shard.efield_tensor(dev)
            assert torch.equal(allf[rank], mine)

    if rank == 0 and not args.no_roofline:
        if args.ordering == "colour":
            if args.mode == "sweep":        # (for rocprofv3: one kind of launch per run)
                # the set-up of the cycle mode -- without cycles --, so that the isolated sweeps run on the working copies a cycle would
                # use: large levels place the blocks their sweeps write while the launch sequences are prepared
                for sc, lr in zip(SC_CYCLE, LR_CYCLE):
                    dev.prepare(sc, lr)
                out["roofline"] = roofline_of(dev, grid, args.workload, sfield if args.source == "dense" else None,
                                              dense_only=args.source == "dense")
            else:
                out["roofline"] = roofline_of(dev, grid, args.workload, None if args.no_dense else sfield)
        reps = 5 if grid.nC <= 128 ** 3 else 3
        rms = dev.time_residual(reps)
        out["residual_kernel"] = {"kernel": dev.last_residual_kernel(), "ms": rms,
                                  "achieved_GBs": RESID_BYTES_PER_CELL * grid.nC / (rms * 1e-3) / 1e9}
    dev.close()

    single = rank == 0 and world == 1 and args.mode == "cycle"
    if single and not args.no_256 and args.workload != "256V":
        # BASELINE.json configs[2]: the 256^3 V-cycle and ITS level-0 sweep, the configuration the north star puts
        # the HBM-roofline target on, measured in the same run
        g2, m2, s2, c2 = build_problem(em, "256V", 1.0)
        v2 = em.VolumeModel(g2, m2, s2)
        var2 = MGParameters(verb=0, cycle=c2, sslsolver=False, linerelaxation=True, semicoarsening=True,
                            vnC=g2.vnC, ordering=args.ordering)
        d2 = DeviceMG(g2, v2, s2.dtype, device=local_rank)
        d2.set_params(var2); d2.set_sfield(s2); d2.set_efield(None)
        ref2 = d2.sfield_norm()
        for sc, lr in zip(SC_CYCLE, LR_CYCLE):
            d2.prepare(sc, lr)
        nw = d2.cycles(3, SC_CYCLE, LR_CYCLE)
        d2._lib.emg3d_mg_sync(d2._h)
        t0 = time.perf_counter()
        n2 = d2.cycles(3, SC_CYCLE, LR_CYCLE)
        d2._lib.emg3d_mg_sync(d2._h)
        t2 = (time.perf_counter() - t0) / 3
        r2 = roofline_of(d2, g2, "256V", None if args.no_dense else s2)
        rms2 = d2.time_residual(3)
        out["config_256V"] = {"workload": "256x256x256 stretched grid, tri-axial anisotropy, V-cycle, "
                                          "semicoarsening+linerelaxation, 1 Hz",
                              "Mcells_per_s": g2.nC / t2 / 1e6, "ms_per_cycle": 1e3 * t2, "roofline": r2,
                              "cycle_algorithmic": {"bytes_per_cycle": cycle_alg_bytes(g2.vnC, c2),
                                                    "frac": cycle_alg_bytes(g2.vnC, c2) / t2 / 1e9 / HBM_PEAK_GBS,
                                                    "frac_executed": cycle_alg_bytes(g2.vnC, c2, executed=(args.ordering == "colour")) / t2 / 1e9 / HBM_PEAK_GBS},
                              "residual_kernel": {"kernel": d2.last_residual_kernel(), "ms": rms2,
                                                  "achieved_GBs": RESID_BYTES_PER_CELL * g2.nC / (rms2 * 1e-3) / 1e9},
                              "rel_error_after": [float(x / ref2) for x in np.r_[nw, n2]],
                              "device_GB": d2.device_bytes / 1e9}
        d2.close()

    batches = [int(x) for x in str(args.batch).split(",") if int(x) > 1]
    if single and batches:
        # Several SOURCES of one frequency through the same launches (solver.solve_sources): one handle, arrays
        # [system][nE]; every system gets bit for bit the arithmetic of a solve of its own (tests/test_gpu_batch.py).
        # Reported beside `value`, never inside it.
        rng = np.random.default_rng(1)
        out["batched_sources"] = []
        for nb, tune in [(nb_, t_) for nb_ in batches for t_ in ((0, 1) if args.batch_tune else (0,))]:
            if nb * grid.nC > 40 * 128 ** 3:
                continue
            # tune = 1: EMG3D_BATCH_TUNE (kernel choice by the lines a launch carries; results then agree with
            # stand-alone solves to rounding instead of bit for bit) -- read when the handle is created
            os.environ["EMG3D_BATCH_TUNE"] = str(tune)
            db = DeviceMG(grid, vm, sfield.dtype, device=local_rank)
            os.environ.pop("EMG3D_BATCH_TUNE")
            db.set_params(var)
            db.set_batch(nb)
            for b in range(nb):
                db.select(b)
                db.set_source([rng.uniform(-800, 800), rng.uniform(-800, 800), rng.uniform(-300, 300),
                               rng.uniform(0, 360), rng.uniform(-30, 30)], sfield.smu0)
            for sc, lr in zip(SC_CYCLE, LR_CYCLE):
                db.prepare(sc, lr)
            db.cycles(3, SC_CYCLE, LR_CYCLE)
            db._lib.emg3d_mg_sync(db._h)
            t0 = time.perf_counter()
            nrm = db.cycles(args.steps, SC_CYCLE, LR_CYCLE)
            db._lib.emg3d_mg_sync(db._h)
            tb = (time.perf_counter() - t0) / args.steps
            refs = []
            for b in range(nb):
                db.select(b)
                refs.append(db.sfield_norm())
            sw = db.time_sweep(3, 5) / 4        # one launch (colour) of the level-0 z-line sweep, all systems
            # algorithmic bytes of a batched launch: e r+w and s per system (144 B/cell), eta and zeta once (56 B/cell)
            alg = (144.0 * nb + 56.0) * grid.nC / 4
            out["batched_sources"].append({
                "systems": nb, "batch_tune": tune, "value": nb * grid.nC / tb / 1e6, "unit": "Mcells/s", "ms_per_cycle": 1e3 * tb,
                "ms_per_cycle_per_system": 1e3 * tb / nb, "sweep_kernel": db.last_sweep_kernel(),
                "level0_sweep_launch_ms": sw, "level0_sweep_alg_bytes_per_launch": alg,
                "level0_sweep_frac_of_hbm_peak": alg / (sw * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "rel_error_after_last": [float(x) for x in np.atleast_2d(nrm)[-1] / np.array(refs)],
                "device_GB": db.device_bytes / 1e9})
            db.close()

    if single and args.multi > 1 and grid.nC <= 128 ** 3:
        # Several independent frequencies sharing the GPU (shard.solve_frequencies(concurrent=K)): each has
        # its own handle and stream and is driven by its own host thread; the coarse levels of one cycle
        # leave most SIMDs idle.  Reported beside `value`, never inside it.
        import threading
        hs = []
        for k in range(args.multi):
            gk, mk, sk, ck = build_problem(em, args.workload, FREQS[k % len(FREQS)])
            dk = DeviceMG(gk, em.VolumeModel(gk, mk, sk), sk.dtype, device=local_rank)
            dk.set_params(var); dk.set_sfield(sk); dk.set_efield(None)
            for sc, lr in zip(SC_CYCLE, LR_CYCLE):
                dk.prepare(sc, lr)
            hs.append(dk)
        for _ in range(2):      # first round: warm-up
            th = [threading.Thread(target=h.cycles, args=(args.steps, SC_CYCLE, LR_CYCLE)) for h in hs]
            t0 = time.perf_counter()
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            tm = time.perf_counter() - t0
        out["concurrent_solves"] = {"solves": args.multi, "value": args.multi * grid.nC * args.steps / tm / 1e6,
                                    "unit": "Mcells/s", "ms_per_cycle_round": 1e3 * tm / args.steps,
                                    "note": "aggregate of independent frequencies on ONE GPU, one stream each"}
        for h in hs:
            h.close()

    if single:
        out["hbm_stream"] = stream_rates()
        for r, cells in ((out.get("roofline"), grid.nC), (out.get("config_256V", {}).get("roofline"), 256 ** 3)):
            if not r:
                continue
            if r.get("traffic"):
                r["traffic_rate_GBs"] = r["traffic"] / (r["launch_ms"] * 1e-3) / 1e9     # counted HBM bytes / launch time
                r["traffic_rate_vs_copy"] = r["traffic_rate_GBs"] / out["hbm_stream"]["copy_GBs"]
            formulation_floor(r, cells, out["hbm_stream"]["copy_GBs"])

    if single and out.get("roofline") and out.get("config_256V", {}).get("roofline"):
        # the configuration the north star puts its 40 % on (BASELINE configs[2]: the 256^3 level-0 sweep), at top level
        r2 = out["config_256V"]["roofline"]
        at = {k: r2.get(k) for k in (
            "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "launch_ms", "launch_ms_stats",
            "source", "launch_ms_sparse_source", "frac_sparse_source", "traffic_sparse_source", "rocprof_average",
            "rocprof_average_sparse_source", "vs_profile", "placement_mode", "placement", "alg_bytes_per_launch",
            "formulation_floor_frac", "traffic_rate_GBs")}
        out["roofline"]["at_256V"] = at
        # ... and as an object of its own beside `roofline` (a parser that keeps top-level keys only must not lose it)
        out["roofline_256V"] = dict(at, workload=out["config_256V"]["workload"], ms_per_cycle=out["config_256V"]["ms_per_cycle"],
                                    Mcells_per_s=out["config_256V"]["Mcells_per_s"],
                                    target_frac=0.40, target_met=bool(at["frac"] is not None and at["frac"] >= 0.40),
                                    note="north star: >= 40 % of HBM peak on this launch.  NOT met: an exact line solve with a cached "
                                         "factor, one colour per launch, moves 784-820 B per block against 200 algorithmic "
                                         "(formulation_floor_frac = its ceiling at this box's copy rate; DESIGN 3.2)")

    if single and not args.no_tol and grid.nC <= 128 ** 3:
        out["time_to_tol"] = time_to_tol(em, args.workload)
        # The price of `value`: it is measured in the colour ordering (the north star's "plane colouring for concurrency"), which
        # agrees with the reference to the solver tolerance.  The reference's own lexicographic order -- the mode in which the
        # per-cycle norms agree to 1e-10 -- runs ~3 n dependent launches per sweep: its cycle time and time to tolerance are
        # put right behind `value` (first screen of the line), not only inside `time_to_tol`.
        lex, col = out["time_to_tol"]["lex"], out["time_to_tol"]["colour"]
        ref = {"ms_per_cycle": lex["ms_per_cycle"], "Mcells_per_s": (grid.nC / (lex["ms_per_cycle"] * 1e-3) / 1e6) if lex["ms_per_cycle"] else None,
               "cycles_to_tol": lex["cycles_to_tol"], "s_to_tol": lex["s_to_tol"],
               "colour_cycles_to_tol": col["cycles_to_tol"], "colour_s_to_tol": col["s_to_tol"], "tol": out["time_to_tol"]["tol"]}
        head = {}
        for k_, v_ in out.items():
            head[k_] = v_
            if k_ == "unit":
                head["reference_order_lex"] = ref
        out = head

    if single and not args.no_tol:
        out["parity"] = parity_16(em)

    if single and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(em, args.workload if grid.nC <= 128 ** 3 else "128F")

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL's version banner sits in the C library's stdout buffer until exit: push it out first, so that
        # the JSON line is the last line on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
