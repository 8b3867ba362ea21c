"""Randomised parity sweep of the HIP path against the oracle (developer aid, not part of the test suite):
random grid shapes (2 ... 48 cells per axis, incl. 3*2^k, 5*2^k, odd), stretched widths, random tri-axial / VTI / isotropic
models with and without mu_r, frequency or Laplace domain, every cycle type, semicoarsening / line relaxation digits,
nu_* settings, clevel caps, both orderings; plus the batched path (two sources) against the single one.
    python tests/tools/fuzz_parity.py [n_cases] [seed]
Prints one line per case and a summary of the worst deviations; exits non-zero on a failure."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emg3d_amd as em                      # noqa: E402
from emg3d_amd.solver import solve_sources  # noqa: E402
from oracle import oracle as orc            # noqa: E402

orc.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
SIZES = [int(x) for x in os.environ.get('FUZZ_SIZES', '2,3,4,5,6,8,10,12,16,20,24,32,40,48').split(',')]
MAXCELLS = int(os.environ.get('FUZZ_MAXCELLS', 40000))
worst = {"field": 0.0, "norm": 0.0}
fails = 0
t_all = time.time()
for case in range(n_cases):
    rng = np.random.default_rng([seed, case])        # every case has its own stream: FUZZ_ONLY=<case> replays it alone
    while True:
        shape = [int(rng.choice(SIZES)) for _ in range(3)]
        if np.prod(shape) <= MAXCELLS and max(shape) >= int(os.environ.get('FUZZ_MINMAX', 4)):
            break
    h = [rng.uniform(20, 60) * rng.uniform(1.0, 1.3) ** np.abs(np.arange(n) - n / 2 + 0.5) for n in shape]
    grid = em.TensorMesh(h, origin=tuple(-hh.sum() / 2 for hh in h))
    rho = 10 ** rng.uniform(-0.5, 2.0, grid.nC)
    aniso = int(rng.integers(0, 4))
    kwm = {}
    if aniso in (1, 3):
        kwm['property_y'] = rho * rng.uniform(1, 3)
    if aniso in (2, 3):
        kwm['property_z'] = rho * rng.uniform(1, 3)
    if rng.random() < 0.3:
        kwm['mu_r'] = rng.uniform(0.8, 1.5, grid.nC)
    model = em.Model(grid, rho, **kwm)
    freq = float(rng.choice([0.1, 1.0, 7.0, -0.5, -3.0]))
    ext = [hh.sum() / 5 for hh in h]
    src = [rng.uniform(-e_, e_) for e_ in ext] + [rng.uniform(0, 360), rng.uniform(-90, 90)]
    sfield = em.get_source_field(grid, src, freq)
    opts = dict(cycle=str(rng.choice(['F', 'V', 'W'])),
                semicoarsening=[False, True, 1, 2, 3, 12, 231][int(rng.integers(0, 7))],
                linerelaxation=[False, True, 1, 4, 7, 56, 123][int(rng.integers(0, 7))],
                nu_init=int(rng.integers(0, 3)), nu_pre=int(rng.integers(0, 3)), nu_coarse=int(rng.integers(1, 3)),
                nu_post=int(rng.integers(1, 3)), maxit=3, tol=1e-12)
    if rng.random() < 0.25:
        opts['clevel'] = int(rng.integers(0, 3))
    ordering = str(rng.choice(['colour', 'lex']))
    # kernel-selection overrides (read when a handle is created): the launch heuristics would otherwise send all of these
    # small grids to the scan kernels
    ENVS = [{}, {}, {"EMG3D_QPL": "0"}, {"EMG3D_QPL": "0", "EMG3D_Q": "2"}, {"EMG3D_QPL": "0", "EMG3D_QM": "1"},
            {"EMG3D_QPL": "0", "EMG3D_THM": "0"}, {"EMG3D_QPL": "0", "EMG3D_TH": "0"}, {"EMG3D_QPL": "0", "EMG3D_TWIST": "0"},
            {"EMG3D_QPL": "0", "EMG3D_SPLIT": "1"}, {"EMG3D_SWEEP": "tpl"}, {"EMG3D_QPL": "0", "EMG3D_TH_LPW": "12"},
            {"EMG3D_QPL": "0", "EMG3D_XT_MIN": "1"}, {"EMG3D_QPL_M2": "2"}, {"EMG3D_GRAPH": "0"}, {"EMG3D_QPL": "5"}]
    env = ENVS[int(rng.integers(0, len(ENVS)))]
    for k in [k for k in os.environ if k.startswith("EMG3D_") and k not in ("EMG3D_POOL_GB",)]:
        del os.environ[k]
    os.environ.update(env)
    ssl = bool(rng.random() < 0.2)
    warm = bool(rng.random() < 0.2)
    tag = f"{case:3d} {tuple(shape)!s:14s} f={freq:5.1f} aniso={aniso} mu={'mu_r' in kwm:d} {ordering:6s} " \
          f"{opts['cycle']} sc={opts['semicoarsening']!s:5s} lr={opts['linerelaxation']!s:5s} nu={opts['nu_init']}{opts['nu_pre']}" \
          f"{opts['nu_coarse']}{opts['nu_post']} cl={opts.get('clevel', '-')} {'bicg ' if ssl else ''}{'warm ' if warm else ''}{' '.join(f'{k[6:]}={v}' for k, v in env.items())}"
    only = os.environ.get('FUZZ_ONLY')
    if only and case not in [int(x) for x in only.split(',')]:
        continue
    try:
        if ssl:
            opts.update(sslsolver='bicgstab', maxit=4)
        e0 = oe0 = None
        if warm:
            e0 = em.Field(grid, (rng.standard_normal(grid.nE) * 1e-9).astype(sfield.dtype), freq=freq)
            e0.ensure_pec
            oe0 = np.array(e0).copy()
        if warm:
            info = em.solve(grid, model, sfield, efield=e0, return_info=True, verb=0, ordering=ordering, **opts)
            e = e0
        else:
            e, info = em.solve(grid, model, sfield, return_info=True, verb=0, ordering=ordering, **opts)
        vm = em.VolumeModel(grid, model, sfield)
        oe, oinfo = orc.solve(orc.Mesh(grid.h, grid.origin), orc.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta),
                              np.array(sfield), efield=oe0, order=0 if ordering == 'lex' else 1, **opts)
        fe = float(np.abs(np.array(e) - oe).max() / max(np.abs(oe).max(), 1e-300))      # (a diverged Krylov solve returns zeros)
        n1, n2 = np.asarray(info['error_at_cycle']), np.asarray(oinfo['error_at_cycle'])
        # per-cycle norms: relative deviation, with a floor of 1e-5 of the source norm (a residual that has dropped further
        # carries the cancellation error of s - A e).  On tiny, badly conditioned systems (Laplace domain, strong stretching)
        # a field that agrees to 1e-12 still moves the norm of a small residual by 1e-12 x condition number: the bound here
        # is a sanity bound that separates rounding (<= 1e-5 seen) from logic errors (>= 1e-4 seen); the FIELD is the measure.
        if n1.shape == n2.shape:
            ne = float((np.abs(n1 - n2) / np.maximum(np.abs(n2), 1e-5 * abs(n2[0]))).max())
        else:
            ne = np.inf
        # batched: two sources in one handle == two solves
        src2 = [rng.uniform(-e_, e_) for e_ in ext] + [rng.uniform(0, 360), rng.uniform(-90, 90)]
        if ssl or warm:
            bat = True
            # forced round-1 two-sided kernels (plain, not mirrored elimination: DESIGN 3.1b) reach 1e-9 ... 2e-8 on
            # ill-conditioned lines; a BiCGSTAB run that DIVERGES in both (zero fields returned) is chaotic: the iteration at
            # which the divergence test fires differs
            weak = any(env.get(k) == '0' for k in ('EMG3D_THM', 'EMG3D_TH'))
            both_diverged = ssl and fe == 0.0 and info['exit'] != 0 and oinfo['exit'] != 0     # (warm: both hand e0 back)
            ok = fe < (1e-6 if ssl else 1e-7 if weak else 1e-8) and (ssl or ne < 3e-5 or ((fe < 1e-11 or weak) and ne < 3e-4)) and \
                (both_diverged or (info['it_mg'] == oinfo['it_mg'] and info['it_ssl'] == oinfo['it_ssl']))
            worst['field'] = max(worst['field'], fe)
            print(f"{tag}  field {fe:.1e} norms {ne:.1e} it {info['it_mg']}/{info['it_ssl']} vs {oinfo['it_mg']}/{oinfo['it_ssl']}  {'ok' if ok else 'FAIL'}", flush=True)
            fails += (not ok)
            continue
        efs, infos = solve_sources(grid, model, [src, src2], freq, verb=0, ordering=ordering, **opts)
        e_b = np.array(em.solve(grid, model, em.SourceField(grid, freq=freq), source=(src2, 0), verb=0, ordering=ordering,
                                **opts))
        e_a = np.array(em.solve(grid, model, em.SourceField(grid, freq=freq), source=(src, 0), verb=0, ordering=ordering,
                                **opts))
        bat = bool(np.array_equal(np.array(efs[0]), e_a) and np.array_equal(np.array(efs[1]), e_b))
        # the norm bound is a sanity bound (see above); where the FIELD agrees to 1e-11 a residual in the cancellation floor
        # of a tiny Laplace-domain grid may move by up to 1e-4 of the floor (seen: 7.5e-5 at field 1e-9 ... 3e-12)
        weak = any(env.get(k) == '0' for k in ('EMG3D_THM', 'EMG3D_TH'))
        ok = fe < (1e-7 if weak else 1e-8) and (ne < 3e-5 or ((fe < 1e-11 or weak) and ne < 3e-4)) and bat and info['it_mg'] == oinfo['it_mg']
        worst['field'] = max(worst['field'], fe); worst['norm'] = max(worst['norm'], ne)
        print(f"{tag}  field {fe:.1e} norms {ne:.1e} batch {'==' if bat else '!='}  {'ok' if ok else 'FAIL'}", flush=True)
        if only:
            print('   gpu   ', n1, info['exit_message'])
            print('   oracle', n2, oinfo['exit_message'])
            print('   shape', shape, 'h0', [hh[:3] for hh in h], 'opts', opts)
        fails += (not ok)
    except Exception as ex:          # noqa
        print(f"{tag}  EXCEPTION {type(ex).__name__}: {ex}", flush=True)
        fails += 1
print(f"{n_cases} cases, {fails} failures, worst field {worst['field']:.2e}, worst norm {worst['norm']:.2e}, {time.time()-t_all:.0f} s")
sys.exit(1 if fails else 0)
