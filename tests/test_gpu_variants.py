"""GPU: alternative code paths selected by environment variables stay correct:
thread-per-line sweep kernel (EMG3D_SWEEP=tpl), x-lines without the transposed
working copy (EMG3D_XT=0), parity-split working copies (EMG3D_SPLIT=1), no
skipping of the idempotent colour pass (EMG3D_SKIP_IDEMPOTENT=0), one-sided
factorisation only (EMG3D_TWIST=0), other lines-per-wave settings."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [{"EMG3D_SWEEP": "tpl"}, {"EMG3D_XT": "0"}, {"EMG3D_SPLIT": "1"},
                                 {"EMG3D_SKIP_IDEMPOTENT": "0"}, {"EMG3D_TWIST": "0"}, {"EMG3D_TW_LPW": "6"},
                                 {"EMG3D_LPW": "8", "EMG3D_TWIST": "0"},
                                 {"EMG3D_SWEEP": "tpl", "EMG3D_XT": "0"}])
@pytest.mark.parametrize("ordering", ["lex", "colour"])
def test_variant_matches_oracle(oracle, monkeypatch, env, ordering):
    import emg3d_amd as em
    for k, v in env.items():
        monkeypatch.setenv(k, v)      # read when a handle is created
    g = load_golden("solves_16.npz")
    grid = em.TensorMesh([g['hx'], g['hy'], g['hz']], origin=g['origin'])
    model = em.Model(grid, g['rho_b'], 2 * g['rho_b'], 3 * g['rho_b'])
    sfield = em.get_source_field(grid, g['src'], float(g['freq']))
    e, info = em.solve(grid, model, sfield, return_info=True, ordering=ordering, cycle='F',
                       semicoarsening=True, linerelaxation=True)
    vm = em.VolumeModel(grid, model, sfield)
    oe, oinfo = oracle.solve(oracle.Mesh(grid.h, grid.origin),
                             oracle.VModel(vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta), np.array(sfield),
                             cycle='F', semicoarsening=True, linerelaxation=True,
                             order=0 if ordering == 'lex' else 1)
    assert info['it_mg'] == oinfo['it_mg']
    np.testing.assert_allclose(info['error_at_cycle'], oinfo['error_at_cycle'], rtol=1e-6)
    assert relerr(e, oe) < 1e-9
