"""MI355X-native multigrid cycle for emg3d (drop-in for the hot path only).

``emg3d_amd.core``   stands where ``emg3d.core`` stands (numba kernels -> HIP);
``emg3d_amd.solver`` mirrors ``emg3d.solver`` (solve / multigrid / krylov / ...);
``emg3d_amd.fields``, ``models``, ``meshes`` carry only the container types the
path needs (Field, SourceField, Model, VolumeModel, TensorMesh);
``emg3d_amd.maps.interp3d`` / ``fields.get_receiver_response`` are the receiver extraction.
"""
from emg3d_amd import core, fields, maps, meshes, models, optimize, shard, solver  # noqa
from emg3d_amd.fields import Field, SourceField, get_h_field, get_receiver_response, get_source_field  # noqa
from emg3d_amd.meshes import TensorMesh  # noqa
from emg3d_amd.models import Model, VolumeModel  # noqa
from emg3d_amd.solver import solve, solve_sources  # noqa

__version__ = "0.1.0"
