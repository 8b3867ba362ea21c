"""
Generate tests/golden/hfield.npz by IMPORTING the reference (emg3d v0.17.0, read-only at /root/reference)
in the build container, exactly as make_golden.py does (no-op numba stub; nothing of the reference is
written into this repository: the fixture holds inputs and expected outputs only).

  reg2_*   the reference's own regression pair for fields.get_h_field (tests/test_fields.py:351-362):
           inputs of `reg_2`, its electric `result` and the stored magnetic `hresult`, plus the output of
           get_h_field run here
  res_*    the reference's second check (test_fields.py:366-381): `res` F-cycle result, H without mu_r
           and with mu_r = 2
  mur_c128 / mur_f64   small random stretched grids with random mu_r, frequency and Laplace domain

Run:  python tests/golden/make_hfield_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference, REF  # noqa: E402


def main():
    _import_reference()
    from emg3d import fields, meshes, models
    dat = np.load(os.path.join(REF, "tests", "data", "regression.npz"), allow_pickle=True)
    out = {}

    def grid_of(prefix):
        return meshes.TensorMesh([dat[f'{prefix}>grid>h{c}'] for c in 'xyz'], origin=dat[f'{prefix}>grid>origin'])

    # reg_2: stored golden of the reference
    grid = grid_of('reg_2')
    model = models.Model(grid, *(dat[f'reg_2>model>property_{c}'] for c in 'xyz'))
    freq = float(dat['reg_2>result>freq'])
    e = fields.Field(grid, dat['reg_2>result>field'], freq=freq)
    h = fields.get_h_field(grid, model, e)
    for c in 'xyz':
        out[f'reg2_h{c}'] = dat[f'reg_2>grid>h{c}']
    out.update(reg2_freq=freq, reg2_e=np.array(e), reg2_h_golden=dat['reg_2>hresult>field'], reg2_h_here=np.array(h),
               reg2_smu0=np.array(e.smu0))
    print('reg_2: stored golden vs here', np.abs(h - dat['reg_2>hresult>field']).max() / np.abs(h).max(), h.is_electric)

    # res: F result, without and with mu_r
    grid = grid_of('res')
    px, py, pz = (dat[f'res>model>property_{c}'] for c in 'xyz')
    freq = float(dat['res>sfield>freq'])
    e = fields.Field(grid, dat['res>Fresult>field'], freq=freq)
    for c in 'xyz':
        out[f'res_h{c}'] = dat[f'res>grid>h{c}']
    out.update(res_freq=freq, res_e=np.array(e))
    out['res_h_nomur'] = np.array(fields.get_h_field(grid, models.Model(grid, px, py, pz), e))
    out['res_h_mur1'] = np.array(fields.get_h_field(grid, models.Model(grid, px, py, pz, mu_r=1.), e))
    out['res_h_mur2'] = np.array(fields.get_h_field(grid, models.Model(grid, px, py, pz, mu_r=2.), e))

    # random grids with random mu_r
    for name, dtype, freq, seed in (('mur_c128', np.complex128, 1.3, 11), ('mur_f64', np.float64, -1.3, 12)):
        rng = np.random.default_rng(seed)
        hx, hy, hz = rng.uniform(20, 60, 8), rng.uniform(20, 60, 6), rng.uniform(20, 60, 5)
        grid = meshes.TensorMesh([hx, hy, hz], origin=np.array([-100., 50., -30.]))
        mu_r = rng.uniform(0.8, 1.5, grid.nC)
        model = models.Model(grid, 1., mu_r=mu_r)
        v = rng.standard_normal(grid.nE)
        if dtype == np.complex128:
            v = v + 1j * rng.standard_normal(grid.nE)
        e = fields.Field(grid, v.astype(dtype), freq=freq)
        h = fields.get_h_field(grid, model, e)
        out.update({f'{name}_hx': hx, f'{name}_hy': hy, f'{name}_hz': hz, f'{name}_freq': freq, f'{name}_mu_r': mu_r,
                    f'{name}_e': np.array(e), f'{name}_h': np.array(h), f'{name}_smu0': np.array(e.smu0)})
        print(name, h.dtype, h.shape, h.vnEx, h.vnEy, h.vnEz)

    np.savez_compressed(os.path.join(HERE, 'hfield.npz'), **out)
    print('wrote hfield.npz', os.path.getsize(os.path.join(HERE, 'hfield.npz')), 'bytes')


if __name__ == '__main__':
    main()
