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

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FREQS = [1.0, 0.25, 0.5, 0.75, 1.5, 2.0, 3.0, 4.0]   # rank r -> FREQS[r]

WORKLOADS = {
    # name: (n, ncore, npad, width, (fx, fy, fz), cycle)    SURVEY 8d C2 / C3
    "128F": (128, 64, 32, 50., (1.06, 1.06, 1.07), 'F'),
    "256V": (256, 128, 64, 25., (1.04, 1.04, 1.045), 'V'),
    "64F": (64, 32, 16, 100., (1.12, 1.12, 1.14), 'F'),
    "32F": (32, 16, 8, 200., (1.25, 1.25, 1.3), 'F'),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
SWEEP_BYTES_PER_CELL = 200.0   # e r+w 96, s 48, eta 48, zeta 8 (complex128, tri-axial)
RESID_BYTES_PER_CELL = 200.0
SWEEP_FLOP_PER_CELL = 1500.0   # minimal band LDL^T line sweep, complex128
FP64_PEAK_TFLOPS = 78.6        # MI355X vector FP64


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


def cpu_baseline(em, ordering):
    """The reference's CPU path stand-in (numba is unavailable and the reference
    cannot travel): the oracle's C++ restatement, -O3 -ffast-math, ONE thread
    (the reference is single-threaded, emg3d/core.py:25), in the reference's
    lexicographic order, on a bounded sample: the 64^3 member of the same
    workload family, 7 F-cycles with semicoarsening + line relaxation (~11 s)."""
    from oracle import oracle as orc
    grid, model, sfield, cycle = build_problem(em, "64F", 1.0)
    vm = em.VolumeModel(grid, model, sfield)
    om = orc.Mesh(grid.h, grid.origin)
    ov = orc.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta)
    t0 = time.perf_counter()
    _, info = orc.solve(om, ov, np.array(sfield), cycle=cycle, semicoarsening=True, linerelaxation=True,
                        maxit=7, tol=1e-30, order=0, fast=True)
    wall = time.perf_counter() - t0
    dt = np.diff(info['runtime_at_cycle'])
    return {
        "value": float(grid.nC / dt.mean() / 1e6), "unit": "Mcells/s per cycle", "cores": 1,
        "kind": "port",
        "sample": f"64^3 stretched tri-axial, 7 F-cycles sc+lr, lexicographic order, "
                  f"C++ -O3 -ffast-math single thread ({wall:.1f} s)",
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="128F", choices=list(WORKLOADS))
    ap.add_argument("--ordering", default="colour", choices=["colour", "lex"])
    ap.add_argument("--mode", default="cycle", choices=["cycle", "sweep"],
                    help="'sweep': only the isolated kernel timings (for rocprofv3 agreement)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--also-256", action="store_true", help="add the 256^3 V-cycle roofline config")
    ap.add_argument("--multi", type=int, default=3,
                    help="N=1 only: also report the aggregate rate of this many concurrent solves (other "
                         "frequencies, own handles and streams) on the one GPU; 0 = skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))

    import torch   # first: its HIP runtime is the one the process uses
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("EMG3D_FORCE_DIST") == "1"   # 1-rank RCCL self-test
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import emg3d_amd as em
    from emg3d_amd.solver import DeviceMG, MGParameters

    freq = FREQS[rank % len(FREQS)]
    grid, model, sfield, cycle = build_problem(em, args.workload, freq)
    vm = em.VolumeModel(grid, model, sfield)
    var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True,
                       vnC=grid.vnC, ordering=args.ordering)
    dev = DeviceMG(grid, vm, sfield.dtype, device=local_rank)
    dev.set_params(var)
    dev.set_sfield(sfield)
    dev.set_efield(None)
    l2_refe = dev.sfield_norm()     # on the device; a multi-threaded host BLAS norm stalls the GPU queues later
    sc_cycle, lr_cycle = [1, 2, 3], [4, 5, 6]

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
        for sc, lr in zip(sc_cycle, lr_cycle):
            dev.prepare(sc, lr)
        if args.warmup > 0:
            norms_w = dev.cycles(args.warmup, sc_cycle, lr_cycle)
        sync()
        t_setup = time.perf_counter() - t_setup0
        # continue the rotation where the warm-up stopped
        rot = args.warmup % 3
        sync()
        t0 = time.perf_counter()
        norms = dev.cycles(args.steps, sc_cycle[rot:] + sc_cycle[:rot], lr_cycle[rot:] + lr_cycle[:rot])
        sync()
        t = time.perf_counter() - t0
        tt = torch.tensor([t], device="cuda", dtype=torch.float64)
        if use_dist:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
        ms_per_step = 1e3 * t_max / args.steps
        value = world * grid.nC * args.steps / t_max / 1e6
        out.update({
            "metric": "Mcells/s per multigrid cycle", "value": value, "unit": "Mcells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "c128",
            "data": "synthetic",
            "config": {"workload": f"{grid.vnC[0]}x{grid.vnC[1]}x{grid.vnC[2]} stretched grid, tri-axial "
                                   f"anisotropy, {cycle}-cycle, semicoarsening+linerelaxation, "
                                   f"nu=0/2/1/2, one frequency per GPU (rank0: {freq} Hz)",
                       "ordering": args.ordering, "cells": int(grid.nC)},
            "rel_error_after": [float(x / l2_refe) for x in np.r_[norms_w if args.warmup else [], norms]],
            "setup_plus_warmup_s": t_setup,
            "device_GB": dev.device_bytes / 1e9,
        })
        # final gather of the fields over RCCL/xGMI (outside the timed region)
        if use_dist:
            from emg3d_amd import shard
            e = dev.get_efield()
            torch.cuda.synchronize(); dist.barrier()
            tg = time.perf_counter()
            allf = shard.gather_fields(e)          # ONE all_gather over RCCL/xGMI
            torch.cuda.synchronize()
            out["gather_ms"] = 1e3 * (time.perf_counter() - tg)
            out["gather_bytes_per_rank"] = int(e.nbytes)
            assert len(allf) == world and all(len(a) == 1 for a in allf)

    if rank == 0:
        # dominant kernel: line-smoother substitution sweep, isolated on the
        # level-0 grid, timed with HIP events on the handle's stream.  One sweep
        # = 4 launches of k_line_sweep (one per colour); algorithmic bytes per
        # launch = 200 B/cell * cells / 4.
        reps = 5 if grid.nC <= 128 ** 3 else 3
        if args.ordering == "colour":
            ms = {d: dev.time_sweep(d, reps) for d in (1, 2, 3)}
            launches = 4
            # average duration of ONE launch of the sweep kernel over the level-0 sweeps of
            # all three directions (what `rocprofv3 --kernel-trace --stats` averages in
            # `bench.py --mode sweep`); the x<->y transposition of the x-direction working
            # copy is a separate kernel and is not inside these events.
            launch_ms = sum(ms.values()) / (3 * launches)
            alg = SWEEP_BYTES_PER_CELL * grid.nC / launches
            ach = alg / (launch_ms * 1e-3) / 1e9
            traffic = None
            tj = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tj):      # HBM bytes per launch from rocprofv3 --pmc (see profiles/README.md)
                with open(tj) as fh:
                    tr = json.load(fh)
                traffic = tr.get(args.workload, {}).get("hbm_bytes_per_launch")
            kname = "k_line_sweep_th<c128,3,8>" if grid.nC <= 128 ** 3 else "k_line_sweep_rp<c128,8>"
            # FP64 co-limit: minimal band-LDL^T line sweep = 1.5 kflop per cell
            flops = SWEEP_FLOP_PER_CELL * grid.nC / launches / (launch_ms * 1e-3) / 1e12
            out["roofline"] = {
                "kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "launch_ms": launch_ms, "launches_per_sweep": launches,
                "sweep_ms": {"x": ms[1], "y": ms[2], "z": ms[3]},
                "alg_bytes_per_launch": alg,
                "fp64": {"alg_flop_per_cell": SWEEP_FLOP_PER_CELL, "achieved": flops, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": flops / FP64_PEAK_TFLOPS},
            }
        rms = dev.time_residual(reps)
        out["residual_kernel"] = {"kernel": "k_residual<c128,1>", "ms": rms,
                                  "achieved_GBs": RESID_BYTES_PER_CELL * grid.nC / (rms * 1e-3) / 1e9}
    dev.close()

    if rank == 0 and args.also_256 and args.mode == "cycle":
        g2, m2, s2, c2 = build_problem(em, "256V", 1.0)
        v2 = em.VolumeModel(g2, m2, s2)
        var2 = MGParameters(verb=0, cycle=c2, sslsolver=False, linerelaxation=True, semicoarsening=True,
                            vnC=g2.vnC, ordering=args.ordering)
        d2 = DeviceMG(g2, v2, s2.dtype, device=local_rank)
        d2.set_params(var2); d2.set_sfield(s2); d2.set_efield(None)
        d2.cycles(3, sc_cycle, lr_cycle)
        t0 = time.perf_counter()
        n2 = d2.cycles(3, sc_cycle, lr_cycle)
        t2 = (time.perf_counter() - t0) / 3
        ms2 = {d: d2.time_sweep(d, 3) for d in (1, 2, 3)}
        out["config_256V"] = {"Mcells_per_s": g2.nC / t2 / 1e6, "ms_per_cycle": 1e3 * t2,
                              "sweep_ms": {"x": ms2[1], "y": ms2[2], "z": ms2[3]},
                              "sweep_hbm_frac": SWEEP_BYTES_PER_CELL * g2.nC / (max(ms2.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "rel_error_after": [float(x / np.linalg.norm(s2)) for x in n2],
                              "device_GB": d2.device_bytes / 1e9}
        d2.close()

    if rank == 0 and world == 1 and args.mode == "cycle" and args.multi > 1 and grid.nC <= 128 ** 3:
        # Several independent frequencies sharing the GPU (shard.solve_frequencies(concurrent=K)): each has
        # its own handle and stream and is driven by its own host thread; the coarse levels of one cycle
        # leave most SIMDs idle.  Reported beside `value`, never inside it.
        import threading
        hs = []
        for k in range(args.multi):
            gk, mk, sk, ck = build_problem(em, args.workload, FREQS[k % len(FREQS)])
            dk = DeviceMG(gk, em.VolumeModel(gk, mk, sk), sk.dtype, device=local_rank)
            dk.set_params(var); dk.set_sfield(sk); dk.set_efield(None)
            for sc, lr in zip(sc_cycle, lr_cycle):
                dk.prepare(sc, lr)
            hs.append(dk)
        for _ in range(2):      # first round: warm-up
            th = [threading.Thread(target=h.cycles, args=(args.steps, sc_cycle, lr_cycle)) for h in hs]
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

    if rank == 0 and not args.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(em, args.ordering)

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
