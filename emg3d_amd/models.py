"""Model containers on the boundary of the hot path: ``Model`` (resistivities
per cell) and ``VolumeModel`` (eta = s mu_0 V sigma, zeta = V / mu_r; reference
emg3d/models.py:554-658), whose arrays are the kernel operands."""
import numpy as np
from scipy.constants import epsilon_0

__all__ = ['Model', 'VolumeModel']


class Model:
    """Resistivity model (Ohm m) on a TensorMesh; optional anisotropy.

    case 0 isotropic, 1 HTI (property_y set), 2 VTI (property_z set), 3 tri-axial
    (reference emg3d/models.py:115-128).
    """

    def __init__(self, grid, property_x=1., property_y=None, property_z=None, mu_r=None,
                 epsilon_r=None, mapping='Resistivity'):
        if mapping not in ('Resistivity', 'Conductivity'):
            raise ValueError("Only 'Resistivity' and 'Conductivity' mappings are supported.")
        self.mapping = mapping
        self.vnC = tuple(grid.vnC)
        self.nC = int(grid.nC)
        self.property_x = self._check(property_x, 'property_x')
        self.property_y = None if property_y is None else self._check(property_y, 'property_y')
        self.property_z = None if property_z is None else self._check(property_z, 'property_z')
        self.mu_r = None if mu_r is None else self._check(mu_r, 'mu_r')
        self.epsilon_r = None if epsilon_r is None else self._check(epsilon_r, 'epsilon_r')
        self.case = (1 if self.property_y is not None else 0) + (2 if self.property_z is not None else 0)
        self.case_names = ['isotropic', 'HTI', 'VTI', 'tri-axial']

    def _check(self, value, name):
        value = np.asarray(value, dtype=np.float64)
        if value.size not in (1, self.nC):
            raise ValueError(f"Shape of {name} must be (), {self.vnC}, or {self.nC}.\n"
                             f"Provided: {value.shape}.")
        if not np.all(np.isfinite(value)) or np.any(value <= 0):
            raise ValueError(f"`{name}` must be all finite and positive.")
        if value.size == self.nC:
            value = value.reshape(self.vnC, order='F')
        return value

    def conductivity(self, name):
        p = getattr(self, name)
        return 1.0 / p if self.mapping == 'Resistivity' else p

    def __repr__(self):
        return f"Model [{self.mapping}]; {self.case_names[self.case]}; {self.vnC}"


def sigma_volume(grid, model):
    """Frequency-independent part of ``VolumeModel``: ``(sv_x, sv_y, sv_z, zeta)`` with
    ``sv = conductivity * cell volume`` (real, F-ordered; ``sv_y``/``sv_z`` alias ``sv_x`` where
    the model does, as ``eta`` does in the reference, models.py:610-624) and
    ``zeta = volume / mu_r``.  ``eta = s mu_0 * sv`` (models.py:631-658) is then one scalar away:
    the ranks of a frequency shard upload ``sv`` and form ``eta`` on the device
    (``DeviceMG.from_sigma_volume``).  Not available with ``epsilon_r`` (eta is then not
    proportional to ``s``)."""
    if model.epsilon_r is not None:
        raise ValueError("sigma_volume: models with epsilon_r are not proportional to s*mu_0")
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    sv_x = np.asfortranarray(vol * model.conductivity('property_x'), dtype=np.float64)
    sv_y = np.asfortranarray(vol * model.conductivity('property_y'), dtype=np.float64) if model.case in (1, 3) else sv_x
    sv_z = np.asfortranarray(vol * model.conductivity('property_z'), dtype=np.float64) if model.case in (2, 3) else sv_x
    zeta = vol if model.mu_r is None else vol / model.mu_r
    return sv_x, sv_y, sv_z, np.asfortranarray(zeta, dtype=np.float64)


class ModelParts(tuple):
    """The tuple ``model_parts`` returns, plus ``epsilon_r`` (F-ordered array or None)."""
    epsilon_r = None


def seps0_of(sval):
    """``s eps_0`` as the device takes it (``emg3d_mg_create_vse``): the imaginary part for a frequency (s = i omega: NumPy's
    ``sval * epsilon_0`` has an exactly-zero real part), the value itself in the Laplace domain."""
    return float(np.imag(sval)) * epsilon_0 if np.iscomplexobj(sval) else float(sval) * epsilon_0


def model_parts(grid, model, raw=False):
    """``(sigma_x, sigma_y, sigma_z, vol, zeta)`` -- the frequency-independent arrays from which the device forms
    ``eta = (s mu_0 V) sigma`` exactly as :class:`VolumeModel` rounds it (``DeviceMG.from_model_parts``,
    ``emg3d_mg_create_vs``); with ``epsilon_r`` the result carries it as attribute ``.epsilon_r`` and the device forms
    ``eta = (s mu_0 V) (sigma - s eps_0 eps_r)`` (``emg3d_mg_create_vse``, reference models.py:639-647).  ``sigma_y`` /
    ``sigma_z`` alias ``sigma_x`` where the model does (reference models.py:610-624).

    ``raw=True``: the model's property arrays as they are plus a flag, ``(p_x, p_y, p_z, vol, zeta, resistivity)`` -- for the
    'Resistivity' mapping the device then takes the reciprocal itself (``from_model_parts(..., resistivity=True)``; an
    IEEE division, the bits of ``Model.conductivity``), which saves three host passes over the model per solve."""
    vol = np.asfortranarray(grid.cell_volumes.reshape(grid.vnC, order='F'), dtype=np.float64)
    get = (lambda name: getattr(model, name)) if raw else model.conductivity
    sx = np.asfortranarray(np.broadcast_to(get('property_x'), grid.vnC), dtype=np.float64)
    sy = np.asfortranarray(np.broadcast_to(get('property_y'), grid.vnC), dtype=np.float64) if model.case in (1, 3) else sx
    sz = np.asfortranarray(np.broadcast_to(get('property_z'), grid.vnC), dtype=np.float64) if model.case in (2, 3) else sx
    zeta = vol if model.mu_r is None else vol / model.mu_r
    out = (sx, sy, sz, vol, np.asfortranarray(zeta, dtype=np.float64))
    out = ModelParts(out + (model.mapping == 'Resistivity',) if raw else out)
    if model.epsilon_r is not None:
        out.epsilon_r = np.asfortranarray(np.broadcast_to(model.epsilon_r, grid.vnC), dtype=np.float64)
    return out


def eta_factored(grid, model, sfield):
    """``(sv_x, sv_y, sv_z, zeta, alpha)`` with REAL arrays ``sv`` such that ``alpha * sv`` is bit for bit
    the ``eta`` of :class:`VolumeModel` -- or ``None`` where that is not possible (epsilon_r).

    ``VolumeModel`` evaluates ``eta = (smu0 * V) * sigma`` (reference models.py:631-658).  In the frequency
    domain ``smu0 = i b`` is purely imaginary and a complex x real product rounds the two parts separately,
    so ``eta = i * ((b * V) * sigma)`` exactly; in the Laplace domain everything is real (``alpha = 1``).
    The device forms ``eta`` from the real array (``emg3d_mg_create_sv``): half the upload, no complex
    temporaries on the host."""
    if model.epsilon_r is not None:
        return None
    smu0 = sfield.smu0
    if np.iscomplexobj(smu0):
        if np.real(smu0) != 0.0:
            return None
        scale, alpha = float(np.imag(smu0)), 1j
    else:
        scale, alpha = float(smu0), 1.0
    vol = grid.cell_volumes.reshape(grid.vnC, order='F')
    sv0 = scale * vol
    sv_x = np.asfortranarray(sv0 * model.conductivity('property_x'))
    sv_y = np.asfortranarray(sv0 * model.conductivity('property_y')) if model.case in (1, 3) else sv_x
    sv_z = np.asfortranarray(sv0 * model.conductivity('property_z')) if model.case in (2, 3) else sv_x
    zeta = vol if model.mu_r is None else vol / model.mu_r
    return sv_x, sv_y, sv_z, np.asfortranarray(zeta, dtype=np.float64), alpha


class VolumeModel:
    """Volume-averaged model arrays (F-ordered ``(nCx, nCy, nCz)``)."""

    def __init__(self, grid, model, sfield):
        self.case = model.case
        vol = grid.cell_volumes.reshape(grid.vnC, order='F')
        self._eta_x = self._eta(vol, model, 'property_x', sfield)
        self._eta_y = self._eta(vol, model, 'property_y', sfield) if self.case in (1, 3) else None
        self._eta_z = self._eta(vol, model, 'property_z', sfield) if self.case in (2, 3) else None
        self._zeta = vol if model.mu_r is None else vol / model.mu_r
        self._zeta = np.asfortranarray(self._zeta, dtype=np.float64)

    @staticmethod
    def _eta(vol, model, name, sfield):
        eta = sfield.smu0 * vol
        sig = model.conductivity(name)
        if model.epsilon_r is None:
            eta = eta * sig
        else:
            eta = eta * (sig - sfield.sval * epsilon_0 * model.epsilon_r)
        return np.asfortranarray(eta)

    @property
    def eta_x(self):
        return self._eta_x

    @property
    def eta_y(self):
        return self._eta_y if self.case in (1, 3) else self._eta_x

    @property
    def eta_z(self):
        return self._eta_z if self.case in (2, 3) else self._eta_x

    @property
    def zeta(self):
        return self._zeta
