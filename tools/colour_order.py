"""Lab experiment: cycles to tolerance for every visiting order of the four line colours (EMG3D_COLOUR_ORDER, lab build),
on the bench problems and on random-resistivity models / other sources / frequencies.
python tools/colour_order.py [forward[:backward] ...]"""
import itertools, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import emg3d_amd as em
from emg3d_amd import _lib
import bench

_lib.use(_lib.LAB_PATH)


def problems():
    for wl in ("128F", "256V"):
        grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
        yield wl, grid, model, sfield, cycle
    # the same 128^3 grid: other frequency and source; a random-resistivity model
    grid, model, _, _ = bench.build_problem(em, "128F", 1.0)
    yield "128F 0.1 Hz, x-dipole off centre", grid, model, em.get_source_field(grid, [300., -200., -400., 0., 0.], 0.1), 'F'
    yield "128V 5 Hz, vertical dipole", grid, model, em.get_source_field(grid, [0., 0., -500., 0., 90.], 5.0), 'V'
    rng = np.random.default_rng(1234)
    rho = 10 ** rng.uniform(-0.5, 1.5, grid.nC)
    yield "128F random resistivities", grid, em.Model(grid, rho, 2 * rho, 3 * rho), em.get_source_field(grid, [0., 0., 0., 30., 10.], 1.0), 'F'
    h = [em.meshes.stretched_widths(48, 24, 40., f) for f in (1.08, 1.05, 1.1)]
    g2 = em.TensorMesh([h[0], h[1][:80], h[2][:64]], origin=(-h[0].sum() / 2, -h[1][:80].sum() / 2, -h[2][:64].sum() / 2))
    rho = 10 ** rng.uniform(0., 2., g2.nC)
    yield "96x80x64 random, isotropic", g2, em.Model(g2, rho), em.get_source_field(g2, [0., 0., 0., 45., 0.], 2.0), 'F'


orders = ["".join(p) for p in itertools.permutations("0123")]
if len(sys.argv) > 1:       # candidate list: forward[:backward] ... (backward as visited; default: forward reversed)
    orders = sys.argv[1:]
table = {}
for name, grid, model, sfield, cycle in problems():
    for p in orders:
        os.environ["EMG3D_COLOUR_ORDER"] = p.split(":")[0]
        os.environ.pop("EMG3D_COLOUR_ORDER_B", None)
        if ":" in p:
            os.environ["EMG3D_COLOUR_ORDER_B"] = p.split(":")[1]
        e, info = em.solve(grid, model, sfield, cycle=cycle, semicoarsening=True, linerelaxation=True, verb=0,
                           return_info=True, tol=1e-6, maxit=40)
        err = np.array(info['error_at_cycle']) / info['ref_error']
        rate = (err[-1] / err[1]) ** (1.0 / max(len(err) - 2, 1))
        table.setdefault(p, []).append((info['it_mg'], rate))
    print(name, flush=True)
    for p in orders:
        it, rate = table[p][-1]
        print(f"   order {p}: {it:2d} cycles, mean reduction per cycle {rate:.3f}", flush=True)
print("\norder: cycles over the problems | geometric mean of the reductions")
for p in orders:
    its = [t[0] for t in table[p]]
    gm = float(np.exp(np.mean(np.log([t[1] for t in table[p]]))))
    print(f"{p}: {its}  {gm:.3f}")
