"""Chain form of the scan kernel (EMG3D_QPL_CHAIN, lab) against the scan form: one colour sweep on grids of short lines, element-wise."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ["EMG3D_HIP_LIB"] = os.path.join(os.getcwd(), "emg3d_amd", "libemg3d_hip_lab.so")
    import emg3d_amd as em
    from emg3d_amd import core
    shape = tuple(int(x) for x in sys.argv[2:5]); direction = int(sys.argv[5])
    rng = np.random.default_rng(3)
    h = [rng.uniform(20., 40., n) for n in shape]
    grid = em.TensorMesh(h, origin=(0., 0., 0.))
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    smu0 = em.SourceField(grid, freq=1.0).smu0
    eta = [np.asfortranarray(smu0 * vol * 10 ** rng.uniform(-1, 1, shape)) for _ in range(3)]
    zeta = np.asfortranarray(vol)
    e = em.Field(grid, rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE), freq=1.0); e.ensure_pec
    s = em.SourceField(grid, (rng.standard_normal(grid.nE) + 1j * rng.standard_normal(grid.nE)) * 1e-6, freq=1.0)
    core._gs(direction, e.fx, e.fy, e.fz, s.fx, s.fy, s.fz, eta[0], eta[1], eta[2], zeta, h[0], h[1], h[2], 1, order=1)
    np.save(sys.argv[6], np.asarray(e))
    sys.exit(0)
for shape, d in (((128, 4, 4), 2), ((128, 4, 4), 3), ((64, 8, 8), 2), ((6, 4, 4), 1)):
    outs = {}
    for c in ("0", "8"):
        f = f"/tmp/chain_{c}.npy"
        subprocess.run([sys.executable, __file__, "child", *map(str, shape), str(d), f], check=True, env=dict(os.environ, EMG3D_QPL_CHAIN=c, EMG3D_LOG="1"),
                       stderr=open(f"/tmp/chain_{c}.err", "w"))
        outs[c] = np.load(f)
    a, b = outs["0"], outs["8"]
    err = np.abs(a - b) / np.abs(a).max()
    bad = np.nonzero(err > 1e-10)[0]
    print(shape, "dir", d, "max rel dev", err.max(), "bad entries", bad.size, "of", a.size, "first", bad[:12])
    print(open("/tmp/chain_8.err").read().splitlines()[:3])
