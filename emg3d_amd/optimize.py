"""Misfit and adjoint-state gradient of ONE (source, frequency) pair on its computational grid -- what
``emg3d.optimize.misfit`` / ``gradient`` (reference emg3d/optimize.py:36-217) and ``Simulation._get_rfield`` /
``_get_bfields`` (emg3d/simulations.py:1131-1213) compute per pair, without the ``Simulation`` / ``Survey``
containers (xarray; out of scope).  Everything field-sized stays in HBM on ONE handle:

    source (built on the device) -> forward solve -> receiver responses (16 B each come back) -> residuals and
    misfit (host scalars) -> residual source (receivers as sources, built on the device) -> back-propagation solve
    -> gradient kernel (-Re(lambda E s mu_0), edges -> cells) -> nC doubles come back.

The loop over (source, frequency) pairs and the mapping of the gradient to the model grid (``maps.grid2grid``)
belong to the caller, as in the reference.
"""
import numpy as np

from . import fields, models, solver


def misfit(synthetic, observed, weights):
    """Weighted least-squares data misfit ``sum(w |syn - obs|^2) / 2`` and the residual (reference
    emg3d/optimize.py:100-111)."""
    residual = np.asarray(synthetic) - np.asarray(observed)
    # (the reference sums xarray DataArrays, which skip NaN entries -- receivers outside the grid or missing data)
    return float(np.nansum(weights * (residual.conj() * residual)).real / 2), residual


def gradient(grid, model, src, freq, rec, observed, weights=None, strength=0, device=0, electric=True, **solver_opts):
    """Misfit and adjoint-state gradient with respect to conductivity for one source and frequency (isotropic
    models without ``epsilon_r`` / ``mu_r``: the reference's limitations, optimize.py:160-170).  ``electric=False``:
    magnetic receivers -- the data are responses of ``H = get_h_field(E)``, the residual sources magnetic point dipoles
    (square loops) with one more division by ``s mu_0`` (simulations.py:1190-1197).

    ``rec = (x, y, z, azimuth, dip)`` point receivers, ``observed`` their data, ``weights`` the data weights
    (default 1).  Returns ``(misfit, grad, info)``: ``grad`` has shape ``grid.vnC`` (Equation (10) of Plessix &
    Mulder 2008 on the computational grid, optimize.py:176-199; NaN receivers are skipped as in
    simulations.py:1181-1183), ``info`` holds the synthetic data and the two solver info dicts.

    SIGN AND CHAIN RULE: ``grad`` is the reference's ``gradient_model`` BEFORE its last two steps, i.e. the sum
    ``grad_x + grad_y + grad_z`` of optimize.py:199.  The reference then maps ``-grad`` to the model grid
    (``maps.grid2grid``, optimize.py:202-211) and applies the property map's ``derivative_chain`` (optimize.py:214):
    the derivative of the misfit with respect to CONDUCTIVITY on this grid is ``-grad`` (what the finite-difference check
    of tests/test_gpu_gradient.py compares with); for a model in another property (resistivity, log-conductivity) the
    caller applies that map's chain factor, as the reference does.  ``model_gradient()`` below returns that quantity."""
    if getattr(model, 'case', 0) != 0:
        raise NotImplementedError("Gradient only implemented for isotropic models.")
    if getattr(model, 'mu_r', None) is not None or getattr(model, 'epsilon_r', None) is not None:
        raise NotImplementedError("Gradient not implemented for el. permittivity / magn. permeability.")
    observed = np.asarray(observed)
    n = observed.size
    weights = np.ones(n) if weights is None else np.broadcast_to(np.asarray(weights), (n,))
    sfield = fields.SourceField(grid, freq=freq)
    smu0 = sfield.smu0
    opts = dict(solver_opts)
    opts.pop('return_info', None)
    # (sigma, V) handle: eta with VolumeModel's rounding, i.e. the fields of solver.solve() bit for bit
    parts = models.model_parts(grid, model, raw=True)
    with solver.DeviceMG.from_model_parts(grid, *parts, smu0=smu0, device=device) as dev:
        # forward field (stays on the device)
        _, finfo = solver.solve(grid, None, sfield, handle=dev, return_info=True, source=(src, strength),
                                download=False, **opts)
        synthetic = dev.get_receiver_response(rec) if electric else dev.get_receiver_response(rec, magnetic=True, smu0=smu0)
        phi, residual = misfit(synthetic, observed, weights)
        dev.vec_alloc(1)
        dev.vec_copy(0, dev.EFIELD)                         # keep the forward field
        # residual source: every receiver becomes a source of strength conj(residual) conj(weight) / s mu_0
        # (simulations.py:1184-1188); magnetic receivers: / s mu_0 once more, loop sources (1190-1197)
        rec = [np.broadcast_to(np.asarray(c, dtype=np.float64), (n,)) for c in rec]
        first = True
        for i in range(n):
            if np.isnan(residual[i]):
                continue
            st = residual[i].conj() * np.conj(weights[i]) / smu0
            if not electric:
                st = st / smu0
            if st == 0:
                continue
            dev.set_source([c[i] for c in rec], smu0, strength=st, accumulate=not first, electric=electric)
            first = False
        if first:
            return phi, np.zeros(grid.vnC, order='F'), dict(synthetic=synthetic, forward=finfo, backward=None)
        rfield = fields.SourceField(grid, freq=freq)
        _, binfo = solver.solve(grid, None, rfield, handle=dev, return_info=True, source='resident',
                                download=False, **opts)
        grad = dev.gradient(0, smu0).reshape(grid.vnC, order='F')
    return phi, grad, dict(synthetic=synthetic, forward=finfo, backward=binfo)


def model_gradient(grid, model, grad):
    """d(misfit) / d(model property) on the computational grid from ``gradient()``'s ``grad``: the reference's last two
    steps without the regridding (optimize.py:201-214 with ``gridding='same'``): the sign, then the chain rule of the
    model's property map -- conductivity: identity; resistivity rho: d sigma / d rho = -1 / rho^2
    (``maps.MapResistivity.derivative_chain``, emg3d/maps.py)."""
    out = -np.asarray(grad)
    mapping = getattr(model, 'mapping', 'Resistivity')
    if mapping == 'Conductivity':
        return out
    if mapping == 'Resistivity':
        rho = np.asarray(model.property_x).reshape(grid.vnC, order='F')
        return out * (-1.0 / rho ** 2)
    raise NotImplementedError(f"model_gradient: property map {mapping!r} (apply its derivative_chain to -grad).")
