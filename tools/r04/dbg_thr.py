import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from types import SimpleNamespace
import emg3d_amd as em
from emg3d_amd.solver import DeviceMG, MGParameters
from oracle import oracle
oracle.build()
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 48, 44), (64, 64, 64), (32, 66, 100), (24, 130, 34)]
for shape in shapes:
    rng = np.random.default_rng(sum(shape))
    h = [rng.uniform(0.5, 2, n) for n in shape]
    grid = em.TensorMesh(h, origin=(0, 0, 0))
    rnd = lambda n: rng.standard_normal(n) + 1j * rng.standard_normal(n)
    eta = [np.asfortranarray(rng.uniform(0.5, 2, shape) * 0.3j) for _ in range(3)]
    zeta = np.asfortranarray(rng.uniform(0.5, 2, shape))
    s = em.Field(grid, rnd(grid.nE), freq=1.)
    var = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC, ordering='colour')
    with DeviceMG(grid, SimpleNamespace(eta_x=eta[0], eta_y=eta[1], eta_z=eta[2], zeta=zeta), np.complex128) as dev:
        dev.set_params(var); dev.set_sfield(s)
        for nu in (1, 2):
            for d in (1, 2, 3):
                e0 = em.Field(grid, rnd(grid.nE), freq=1.)
                dev.set_efield(e0); dev.smooth(nu, d)
                e = dev.get_efield()
                eo = np.array(e0)
                oracle.gauss_seidel(grid.vnC, eo, np.array(s), *eta, zeta, *grid.h, nu, direction=d, order=1)
                err = float(np.abs(np.asarray(e) - eo).max() / np.abs(eo).max())
                msg = ""
                if err > 1e-9:
                    # the component along the line: which indices along the line / which lines are wrong
                    c = ("fx", "fy", "fz")[d - 1]
                    df = np.abs(getattr(em.Field(grid, np.asarray(e), freq=1.), c) - getattr(em.Field(grid, eo, freq=1.), c)) > 1e-9
                    ax = d - 1
                    along = np.where(df.any(axis=tuple(i for i in range(3) if i != ax)))[0]
                    lines = df.any(axis=ax)
                    msg = " wrong L-entries %d of %d; along-line idx %s..%s (n=%d); wrong lines %d of %d, first %s" % (
                        df.sum(), df.size, along.min(), along.max(), shape[ax], lines.sum(), lines.size, np.argwhere(lines)[:4].tolist())
                print(shape, "nu", nu, "dir", d, dev.last_sweep_kernel(), "relerr %.2e" % err, msg, flush=True)
