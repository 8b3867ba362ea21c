import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import emg3d_amd as em
from emg3d_amd.solver import DeviceMG, MGParameters
from emg3d_amd import shard
from oracle import oracle
import bench
from conftest import relerr
from test_gpu_fullsize import _smooth_field

which = sys.argv[1:] or ["sweep", "bic", "shard"]
grid, model, sfield, cycle = bench.build_problem(em, "128F", 1.0)
vm = em.VolumeModel(grid, model, sfield)
if "sweep" in which:
    var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC, ordering='colour')
    e0 = _smooth_field(grid, 7)
    s = em.SourceField(grid, np.array(_smooth_field(grid, 8)) * 1e-3, freq=1.0)
    eta = [np.asfortranarray(a) for a in (vm.eta_x, vm.eta_y, vm.eta_z)]
    zeta = np.asfortranarray(vm.zeta)
    with DeviceMG(grid, vm, np.complex128) as dev:
        dev.set_params(var); dev.set_sfield(s)
        for direction in (1, 2, 3):
            dev.set_efield(e0); dev.smooth(1, direction)
            got = dev.get_efield()
            ref = np.array(e0)
            oracle.gauss_seidel(grid.vnC, ref, np.array(s), *eta, zeta, *grid.h, 1, direction=direction, order=1)
            d = np.abs(got - ref)
            k = int(d.argmax())
            print("sweep dir", direction, dev.last_sweep_kernel(), "relerr", relerr(got, ref), "max at", k, got[k], ref[k],
                  "changed", relerr(got, np.array(e0)), flush=True)
            # by component
            nx, ny, nz = grid.vnC
            off = np.cumsum([0, nx*(ny+1)*(nz+1), (nx+1)*ny*(nz+1), (nx+1)*(ny+1)*nz])
            for c in range(3):
                print("   comp", c, np.abs(got[off[c]:off[c+1]] - ref[off[c]:off[c+1]]).max() / np.abs(ref).max())
if "bic" in which:
    e, info = em.solve(grid, model, sfield, cycle=cycle, sslsolver='bicgstab', semicoarsening=True, linerelaxation=True, return_info=True, tol=1e-6, verb=0)
    print({k: info[k] for k in ('exit', 'exit_message', 'it_ssl', 'it_mg', 'rel_error', 'abs_error', 'ref_error')})
    l2 = oracle.residual(oracle.Mesh(grid.h, grid.origin), oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta), np.array(sfield), np.array(e), True, fast=True)
    print("oracle l2", l2, l2 / info['ref_error'], l2 / info['abs_error'] - 1)
    e2 = em.solve(grid, model, sfield, cycle=cycle, semicoarsening=True, linerelaxation=True, tol=1e-7, verb=0)
    print("vs mg", relerr(np.array(e), np.array(e2)))
if "shard" in which:
    kw = dict(cycle=cycle, semicoarsening=True, linerelaxation=True, tol=1e-6, verb=0)
    src = [0., 0., 0., 30., 10.]
    f = bench.FREQS[3]
    (a, ia), = shard.solve_frequencies(grid, model, src, [f], **kw)
    (b, ib), = shard.solve_frequencies(grid, model, src, [f], **kw)
    print("seq-seq equal", np.array_equal(np.array(a), np.array(b)), ia['it_mg'], ib['it_mg'])
    res = shard.solve_frequencies(grid, model, src, bench.FREQS[:4], concurrent=3, **kw)
    c = res[3][0]
    print("conc-seq equal", np.array_equal(np.array(a), np.array(c)), relerr(np.array(c), np.array(a)), res[3][1]['it_mg'],
          np.abs(res[3][1]['error_at_cycle'] / ia['error_at_cycle'] - 1).max())
    res2 = shard.solve_frequencies(grid, model, src, bench.FREQS[:4], concurrent=1, **kw)
    print("seq4-seq equal", np.array_equal(np.array(a), np.array(res2[3][0])), relerr(np.array(res2[3][0]), np.array(a)))
    (d, idd), = shard.solve_frequencies(grid, model, src, [f], **kw)
    print("seq after equal", np.array_equal(np.array(a), np.array(d)), relerr(np.array(d), np.array(a)))
