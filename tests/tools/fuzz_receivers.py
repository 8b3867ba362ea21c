"""Randomised check of the receiver / source / gradient kernels against the oracle restatements (developer aid):
random grids (3 ... 40 cells per axis, also < 4 interior points: forced linear), random complex and real fields, receivers
inside, on nodes, in the first / last interval, outside; random finite dipoles and paths against the host twin.
    python tests/tools/fuzz_receivers.py [n_cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emg3d_amd as em                      # noqa: E402
from emg3d_amd.solver import DeviceMG       # noqa: E402
from oracle import interp as oi             # noqa: E402
from oracle import gradient as og           # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(rec=0.0, src=0.0, e2c=0.0)
fails = 0
for case in range(n_cases):
    if case and case % 500 == 0:
        print(f"... {case} cases so far, {fails} failures", flush=True)
    shape = [int(rng.integers(3, 41)) for _ in range(3)]
    while np.prod(shape) > 30000:
        shape[int(np.argmax(shape))] = max(shape[int(np.argmax(shape))] // 2, 3)
    h = [rng.uniform(10, 80, n) for n in shape]
    origin = rng.uniform(-500, 0, 3)
    grid = em.TensorMesh(h, origin=tuple(origin))
    cplx = rng.random() < 0.7
    v = rng.standard_normal(grid.nE) + (1j * rng.standard_normal(grid.nE) if cplx else 0)
    field = em.Field(grid, v.astype(np.complex128 if cplx else np.float64), freq=1.0 if cplx else -1.0)
    nodes = (grid.nodes_x, grid.nodes_y, grid.nodes_z)
    nrec = int(rng.integers(1, 30))
    xyz = [rng.uniform(nd[0] - 20, nd[-1] + 20, nrec) for nd in nodes]
    for a in range(3):                       # some receivers exactly on nodes / cell centres
        k = rng.integers(0, nrec, max(nrec // 4, 1))
        xyz[a][k] = rng.choice(nodes[a], k.size)
    rec = (xyz[0], xyz[1], xyz[2], rng.uniform(-180, 180, nrec), rng.uniform(-90, 90, nrec))
    try:
        got = em.get_receiver_response(grid, field, rec)
    except Exception as ex:
        print(case, shape, 'nrec', nrec, cplx, 'EXCEPTION', ex); fails += 1; continue
    ref = oi.get_receiver_response(grid.h, grid.origin, np.array(field), rec)
    ok = np.array_equal(np.isnan(got), np.isnan(ref))
    if not ok:
        k = np.nonzero(np.isnan(got) != np.isnan(ref))[0]
        for kk in k[:3]:
            print("   nan mismatch at receiver", kk, "gpu", got[kk], "oracle", ref[kk], "xyz", [float(x[kk]) for x in xyz],
                  "node index (nearest, distance):", [(int(np.argmin(abs(nd - x[kk]))), float(np.min(abs(nd - x[kk])))) for nd, x in zip(nodes, xyz)],
                  "nN", [nd.size for nd in nodes])
    m = ~np.isnan(ref)
    er = float(np.abs(got[m] - ref[m]).max() / max(np.abs(ref[m]).max(), 1e-300)) if m.any() else 0.0
    # source: device kernel against the host twin (itself pinned to the reference's fixtures)
    ext = [(nd[1] + 1e-3, nd[-2] - 1e-3) for nd in nodes]
    kind = int(rng.integers(0, 3))
    if kind == 0:
        src = [rng.uniform(*ext[0]), rng.uniform(*ext[0]), rng.uniform(*ext[1]), rng.uniform(*ext[1]), rng.uniform(*ext[2]), rng.uniform(*ext[2])]
    elif kind == 1:
        src = [[rng.uniform(*ext[a]) for _ in range(4)] for a in range(3)]
    else:
        src = [np.mean(ext[0]), np.mean(ext[1]), np.mean(ext[2]), rng.uniform(0, 360), rng.uniform(-90, 90)]
    electric = bool(rng.random() < 0.7)
    length = float(min(1.0, 0.2 * min(hh.min() for hh in h)))
    strength = [0, 2.5, 1 - 2j][int(rng.integers(0, 3))] if cplx else [0, 2.5][int(rng.integers(0, 2))]
    import contextlib, io, warnings
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter('ignore')         # ("Normalizing Source": long oblique dipoles, as the reference)
        sf = em.get_source_field(grid, src, 1.0 if cplx else -1.0, strength=strength, electric=electric, length=length)

    class VM:
        eta_x = eta_y = eta_z = np.asfortranarray(np.ones(grid.vnC) * (1j if cplx else 1.0))
        zeta = np.asfortranarray(np.ones(grid.vnC))
    with DeviceMG(grid, VM, sf.dtype) as dev:
        dev.set_source(src, sf.smu0, strength=strength, electric=electric, length=length)
        dsrc = dev.vec_get(dev.SFIELD)
    es = float(np.abs(dsrc - np.array(sf)).max() / max(np.abs(np.array(sf)).max(), 1e-300))
    # edges2cellaverages
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    outs = [np.zeros(grid.vnC, dtype=field.dtype, order='F') for _ in range(3)]
    em.maps.edges2cellaverages(field.fx, field.fy, field.fz, vol, *outs)
    refs = og.edges2cellaverages(field.fx, field.fy, field.fz, vol)
    ee = max(float(np.abs(o - r).max() / np.abs(r).max()) for o, r in zip(outs, refs))
    good = ok and er < 1e-10 and es < 1e-12 and ee < 1e-13
    fails += (not good)
    for k_, v_ in (('rec', er), ('src', es), ('e2c', ee)):
        worst[k_] = max(worst[k_], v_)
    if not good:
        print(f"{case:3d} {tuple(shape)} cplx={cplx} nrec={nrec} kind={kind} electric={electric}: nan-pattern {'ok' if ok else 'DIFF'} rec {er:.1e} src {es:.1e} e2c {ee:.1e} FAIL", flush=True)
print(f"{n_cases} cases, {fails} failures, worst {worst}")
sys.exit(1 if fails else 0)
