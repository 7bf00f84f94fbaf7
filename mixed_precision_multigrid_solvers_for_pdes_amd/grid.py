"""Grid metadata (host side).  Mirrors multigrid.core.grid.Grid (core/grid.py:10-217): same
constructor, attributes, coarsen/refine rules and error messages.  The reference materialises
X, Y, values and residual eagerly (5 n^2 host arrays); here they are built on first use so a
16385^2 grid does not cost 10 GB of host memory before the device is even touched."""
import ctypes as C

import numpy as np

from . import _lib


class Grid:
    def __init__(self, nx, ny, domain=(0.0, 1.0, 0.0, 1.0), dtype=np.float64):
        if nx < 3 or ny < 3:
            raise ValueError("Grid must have at least 3 points in each direction")      # core/grid.py:34-35
        self.nx, self.ny = int(nx), int(ny)
        self.domain = tuple(domain)
        self.dtype = dtype
        self.hx = (domain[1] - domain[0]) / (nx - 1)                                      # core/grid.py:43-45
        self.hy = (domain[3] - domain[2]) / (ny - 1)
        self.h = min(self.hx, self.hy)
        self.x = np.linspace(domain[0], domain[1], nx, dtype=dtype)
        self.y = np.linspace(domain[2], domain[3], ny, dtype=dtype)
        self._X = self._Y = self._values = self._residual = None

    # lazily materialised n^2 arrays -------------------------------------------------------
    def _mesh(self):
        if self._X is None:
            self._X, self._Y = np.meshgrid(self.x, self.y, indexing="ij")                 # core/grid.py:50
        return self._X, self._Y

    @property
    def X(self):
        return self._mesh()[0]

    @property
    def Y(self):
        return self._mesh()[1]

    @property
    def values(self):
        if self._values is None:
            self._values = np.zeros((self.nx, self.ny), dtype=self.dtype)
        return self._values

    @values.setter
    def values(self, v):
        self._values = v

    @property
    def residual(self):
        if self._residual is None:
            self._residual = np.zeros((self.nx, self.ny), dtype=self.dtype)
        return self._residual

    @residual.setter
    def residual(self, v):
        self._residual = v

    @property
    def shape(self):
        return (self.nx, self.ny)

    @property
    def size(self):
        return self.nx * self.ny

    def interior_slice(self):
        return (slice(1, -1), slice(1, -1))

    def boundary_slice(self, side):                                                       # core/grid.py:74-90
        if side == "left":
            return (slice(0, 1), slice(None))
        if side == "right":
            return (slice(-1, None), slice(None))
        if side == "bottom":
            return (slice(None), slice(0, 1))
        if side == "top":
            return (slice(None), slice(-1, None))
        raise ValueError(f"Unknown boundary side: {side}")

    def apply_dirichlet_bc(self, value, side=None):                                       # core/grid.py:92-113
        sides = ["left", "right", "bottom", "top"] if side in (None, "all") else [side]
        for s in sides:
            self.values[self.boundary_slice(s)] = value

    def coarsen(self):                                                                    # core/grid.py:140-157
        if (self.nx - 1) % 2 != 0 or (self.ny - 1) % 2 != 0:
            raise ValueError("Cannot coarsen grid: need even number of interior points")
        return Grid((self.nx - 1) // 2 + 1, (self.ny - 1) // 2 + 1, self.domain, self.dtype)

    def refine(self):                                                                     # core/grid.py:159-172
        return Grid(2 * (self.nx - 1) + 1, 2 * (self.ny - 1) + 1, self.domain, self.dtype)

    def l2_norm(self, field=None):
        """sqrt(hx*hy*sum(field^2)) over all cells (core/grid.py:174-187) -- wave64 shuffle
        reduction on the device, fp64 accumulation."""
        f = _lib.as_c(self.values if field is None else field)
        if f.shape != self.shape:
            raise ValueError(f"Field shape {f.shape} doesn't match grid shape {self.shape}")
        out = C.c_double(0.0)
        _lib.check(_lib.load().mg_op_norm(_lib.dtype_code(f.dtype), self.nx, self.ny, self.hx, self.hy,
                                          _lib.ptr(f), C.byref(out)))
        return out.value

    def max_norm(self, field=None):                                                       # core/grid.py:189-202
        return float(np.max(np.abs(self.values if field is None else field)))

    def copy(self):
        g = Grid(self.nx, self.ny, self.domain, self.dtype)
        g.values = self.values.copy()
        g.residual = self.residual.copy()
        return g

    def __repr__(self):
        return f"Grid(nx={self.nx}, ny={self.ny}, domain={self.domain}, dtype={self.dtype})"

    __str__ = __repr__
