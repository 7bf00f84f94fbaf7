"""Operator plugins.  Mirror multigrid.operators (operators/base.py, laplacian.py, transfer.py):
same class names, constructor arguments, method signatures and error behaviour; the arithmetic runs
in libmghip.so on the GPU."""
import numpy as np

from . import _lib


class BaseOperator:
    def __init__(self, name="BaseOperator"):
        self.name = name

    def __str__(self):
        return self.name


class LaplacianOperator(BaseOperator):
    """A = coefficient * (5-point Laplacian) (operators/laplacian.py:15-124).

    The reference's default coefficient (+1.0) is inconsistent with its own smoothers, which relax
    -Laplace(u) = rhs, and makes its multigrid diverge; coefficient=-1.0 is the
    self-consistent configuration.  The default is kept for signature parity."""

    def __init__(self, coefficient=1.0):
        super().__init__(f"Laplacian(coeff={coefficient})")
        self.coefficient = coefficient

    def can_apply(self, grid):
        return grid.nx >= 3 and grid.ny >= 3

    def _check(self, grid, field):
        if not self.can_apply(grid):
            raise ValueError(f"Cannot apply Laplacian to grid {grid.shape}")
        if field.shape != grid.shape:
            raise ValueError(f"Field shape {field.shape} doesn't match grid shape {grid.shape}")

    def apply(self, grid, field=None):
        f = _lib.as_c(grid.values if field is None else field)
        self._check(grid, f)
        out = np.empty_like(f)
        _lib.check(_lib.load().mg_op_apply(_lib.dtype_code(f.dtype), grid.nx, grid.ny, grid.hx, grid.hy,
                                           float(self.coefficient), _lib.ptr(f), _lib.ptr(out)))
        return out

    def residual(self, grid, u, f):
        u, f = _lib.as_c(u), _lib.as_c(f)
        dt = np.result_type(u.dtype, f.dtype)        # `f - Au` promotes like NumPy does
        u, f = np.ascontiguousarray(u, dtype=dt), np.ascontiguousarray(f, dtype=dt)
        self._check(grid, u)
        self._check(grid, f)
        r = np.empty_like(u)
        _lib.check(_lib.load().mg_op_residual(_lib.dtype_code(u.dtype), grid.nx, grid.ny, grid.hx, grid.hy,
                                              float(self.coefficient), _lib.ptr(u), _lib.ptr(f), _lib.ptr(r)))
        grid.residual = r.copy()                      # side effect kept (operators/laplacian.py:121)
        return r


def _grid_dtype(grid):
    return np.dtype(grid.dtype)


class RestrictionOperator(BaseOperator):
    """Fine -> coarse transfer (operators/transfer.py:15-148).  'full_weighting' is the hot-path
    method and the one implemented on the device; the reference's other two methods
    ('injection', 'half_weighting') are accepted by the constructor and rejected at apply()."""

    def __init__(self, method="full_weighting"):
        super().__init__(f"Restriction({method})")
        if method not in ["injection", "full_weighting", "half_weighting"]:
            raise ValueError(f"Unknown restriction method: {method}")
        self.method = method

    def can_apply(self, fine_grid, coarse_grid):
        return (coarse_grid.nx == (fine_grid.nx - 1) // 2 + 1 and
                coarse_grid.ny == (fine_grid.ny - 1) // 2 + 1)

    def apply(self, fine_grid, field, coarse_grid):
        if not self.can_apply(fine_grid, coarse_grid):
            raise ValueError(f"Cannot restrict from {fine_grid.shape} to {coarse_grid.shape}")
        field = _lib.as_c(field)
        if field.shape != fine_grid.shape:
            raise ValueError(f"Field shape {field.shape} doesn't match fine grid {fine_grid.shape}")
        if self.method != "full_weighting":
            raise NotImplementedError(f"restriction method {self.method!r} is outside the accelerated hot path")
        out = np.empty(coarse_grid.shape, dtype=_grid_dtype(coarse_grid))
        _lib.check(_lib.load().mg_op_restrict_fw(_lib.dtype_code(field.dtype), _lib.dtype_code(out.dtype),
                                                 fine_grid.nx, fine_grid.ny, _lib.ptr(field), _lib.ptr(out)))
        return out


class ProlongationOperator(BaseOperator):
    """Coarse -> fine transfer (operators/transfer.py:151-267); 'bilinear' runs on the device and
    reproduces the reference's far-edge behaviour."""

    def __init__(self, method="bilinear"):
        super().__init__(f"Prolongation({method})")
        if method not in ["injection", "bilinear"]:
            raise ValueError(f"Unknown prolongation method: {method}")
        self.method = method

    def can_apply(self, coarse_grid, fine_grid):
        return (fine_grid.nx == 2 * (coarse_grid.nx - 1) + 1 and
                fine_grid.ny == 2 * (coarse_grid.ny - 1) + 1)

    def apply(self, coarse_grid, field, fine_grid):
        if not self.can_apply(coarse_grid, fine_grid):
            raise ValueError(f"Cannot prolongate from {coarse_grid.shape} to {fine_grid.shape}")
        field = _lib.as_c(field)
        if field.shape != coarse_grid.shape:
            raise ValueError(f"Field shape {field.shape} doesn't match coarse grid {coarse_grid.shape}")
        if self.method != "bilinear":
            raise NotImplementedError(f"prolongation method {self.method!r} is outside the accelerated hot path")
        out = np.empty(fine_grid.shape, dtype=_grid_dtype(fine_grid))
        _lib.check(_lib.load().mg_op_prolong_bilinear(_lib.dtype_code(field.dtype), _lib.dtype_code(out.dtype),
                                                      coarse_grid.nx, coarse_grid.ny, _lib.ptr(field), _lib.ptr(out)))
        return out


class HelmholtzOperator(LaplacianOperator):
    """A u = coefficient * (Laplacian_h u - shift * u): for coefficient = -1 the SPD operator -Laplacian + shift of
    an implicit heat-equation step (applications/heat_equation.py:209-220 builds this system and, lacking a shifted
    multigrid, relaxes it with Gauss-Seidel, :459-497).  shift = 0 is LaplacianOperator(coefficient) bit for bit."""

    def __init__(self, shift, coefficient=-1.0):
        if not shift >= 0:
            raise ValueError("the Helmholtz shift must be >= 0")
        super().__init__(coefficient)
        self.name = f"Helmholtz(coeff={coefficient}, shift={shift})"
        self.shift = float(shift)

    def residual(self, grid, u, f):
        u, f = _lib.as_c(u), _lib.as_c(f)
        dt = np.result_type(u.dtype, f.dtype)
        u, f = np.ascontiguousarray(u, dtype=dt), np.ascontiguousarray(f, dtype=dt)
        self._check(grid, u)
        self._check(grid, f)
        r = np.empty_like(u)
        _lib.check(_lib.load().mg_op_helmholtz(_lib.dtype_code(dt), 0, grid.nx, grid.ny, grid.hx, grid.hy,
                                               float(self.coefficient), self.shift, 1.0, 0, _lib.ptr(u), _lib.ptr(f), _lib.ptr(r)))
        grid.residual = r.copy()
        return r

    def apply(self, grid, field=None):
        u = _lib.as_c(grid.values if field is None else field)
        saved = grid._residual
        out = -self.residual(grid, u, np.zeros_like(u))   # boundary rows / columns 0 like LaplacianOperator.apply
        grid._residual = saved
        return out


class DiffusionOperator(BaseOperator):
    """A u = coefficient * div(a grad u)  (coefficient = -1: the SPD operator -div(a grad u)).

    BASELINE config 5 names a variable-coefficient problem; the reference has no such operator (a README bullet
    only), so this class and its discretisation are ours: `a` is sampled on the grid vertices
    (array of grid.shape, or a callable a(X, Y)), face values are arithmetic means, coarse levels re-discretise with
    `a` injected.  With a == 1 it reproduces LaplacianOperator(coefficient) bit for bit on dyadic grids."""

    def __init__(self, a, coefficient=-1.0):
        super().__init__(f"Diffusion(coeff={coefficient})")
        self.a = a
        self.coefficient = coefficient

    def field(self, grid, dtype=None):
        a = self.a(grid.X, grid.Y) if callable(self.a) else self.a
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), grid.shape), dtype=dtype or np.float64)
        if np.any(a <= 0):
            raise ValueError("the diffusion coefficient must be positive")
        return a

    def can_apply(self, grid):
        return grid.nx >= 3 and grid.ny >= 3

    def residual(self, grid, u, f):
        u, f = _lib.as_c(u), _lib.as_c(f)
        dt = np.result_type(u.dtype, f.dtype)
        u, f = np.ascontiguousarray(u, dtype=dt), np.ascontiguousarray(f, dtype=dt)
        if u.shape != grid.shape or f.shape != grid.shape:
            raise ValueError(f"Field shape {u.shape} doesn't match grid shape {grid.shape}")
        a = self.field(grid, dt)
        r = np.empty_like(u)
        _lib.check(_lib.load().mg_op_residual_var(_lib.dtype_code(dt), grid.nx, grid.ny, grid.hx, grid.hy,
                                                  float(self.coefficient), _lib.ptr(a), _lib.ptr(u), _lib.ptr(f), _lib.ptr(r)))
        grid.residual = r.copy()
        return r

    def apply(self, grid, field=None):
        u = _lib.as_c(grid.values if field is None else field)
        zero = np.zeros_like(u)
        saved = grid._residual
        out = -self.residual(grid, u, zero)          # A u = -(0 - A u); boundary rows/cols are 0 like LaplacianOperator.apply
        grid._residual = saved
        return out
