"""Tensor mesh container (the attributes of ``emg3d.meshes._TensorMesh`` that
the multigrid path reads; reference emg3d/meshes.py:66-147).  Gridding helpers
of the reference (construct_mesh, skin depth, ...) are out of scope."""
import numpy as np


class TensorMesh:
    """Rectilinear mesh defined by cell widths ``h = [hx, hy, hz]`` and origin."""

    def __init__(self, h, origin=(0., 0., 0.)):
        self.h = [np.array(w, dtype=np.float64) for w in h]
        if len(self.h) != 3:
            raise ValueError("Provided grid must be a 3D grid.")
        self.origin = np.array(origin, dtype=np.float64)
        nodes = [np.r_[0., w.cumsum()] + o for w, o in zip(self.h, self.origin)]
        self.nodes_x, self.nodes_y, self.nodes_z = nodes
        cc = [(n[1:] + n[:-1]) / 2 for n in nodes]
        self.cell_centers_x, self.cell_centers_y, self.cell_centers_z = cc

        self.shape_cells = tuple(int(w.size) for w in self.h)
        self.shape_nodes = tuple(n + 1 for n in self.shape_cells)
        nx, ny, nz = self.shape_cells
        self.shape_edges_x = (nx, ny + 1, nz + 1)
        self.shape_edges_y = (nx + 1, ny, nz + 1)
        self.shape_edges_z = (nx + 1, ny + 1, nz)
        self.n_cells = nx * ny * nz
        self.n_nodes = int(np.prod(self.shape_nodes))
        self.n_edges_x = int(np.prod(self.shape_edges_x))
        self.n_edges_y = int(np.prod(self.shape_edges_y))
        self.n_edges_z = int(np.prod(self.shape_edges_z))
        self.n_edges_per_direction = (self.n_edges_x, self.n_edges_y, self.n_edges_z)
        self.n_edges = sum(self.n_edges_per_direction)

        # short aliases used throughout the reference
        self.x0 = self.origin
        self.vnC, self.nC = self.shape_cells, self.n_cells
        self.vnN, self.nN = self.shape_nodes, self.n_nodes
        self.vnEx, self.vnEy, self.vnEz = self.shape_edges_x, self.shape_edges_y, self.shape_edges_z
        self.nEx, self.nEy, self.nEz = self.n_edges_per_direction
        self.vnE, self.nE = self.n_edges_per_direction, self.n_edges
        self._vol = None

    def __repr__(self):
        nx, ny, nz = self.shape_cells
        return f"TensorMesh: {nx} x {ny} x {nz} ({self.n_cells:,})"

    @property
    def cell_volumes(self):
        """Cell volumes as 1-D array, x fastest (reference meshes.py:140-147)."""
        if self._vol is None:
            self._vol = (self.h[0][None, None, :] * self.h[1][None, :, None] *
                         self.h[2][:, None, None]).ravel()
        return self._vol


_TensorMesh = TensorMesh


def stretched_widths(ncore, npad, width, factor):
    """Widths [w f^npad .. w f, w x ncore, w f .. w f^npad] used by the synthetic
    benchmark grids."""
    pad = width * float(factor) ** np.arange(1, npad + 1)
    return np.r_[pad[::-1], np.full(ncore, float(width)), pad]
