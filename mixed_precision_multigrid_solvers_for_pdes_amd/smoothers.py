"""Smoother plugins.  Mirror multigrid.solvers.base / smoothers / iterative: ConvergenceHistory,
BaseSolver, IterativeSolver, JacobiSmoother, WeightedJacobiSmoother, GaussSeidelSmoother,
EnhancedJacobiSolver -- same constructors and smooth()/solve() signatures; sweeps run on the GPU."""
import ctypes as C
import logging
import time

import numpy as np

from . import _lib

logger = logging.getLogger(__name__)


class ConvergenceHistory:                                                    # solvers/base.py:17-64
    def __init__(self):
        self.residual_norms, self.iteration_times = [], []
        self.precision_levels, self.grid_levels = [], []

    def record_iteration(self, residual_norm, iteration_time, precision_level, grid_level=None):
        self.residual_norms.append(residual_norm)
        self.iteration_times.append(iteration_time)
        self.precision_levels.append(precision_level)
        self.grid_levels.append(grid_level)

    def get_convergence_rate(self):
        if len(self.residual_norms) < 3:
            return 0.0
        recent = self.residual_norms[-5:]
        ratios = []
        for i in range(1, len(recent)):
            if recent[i - 1] > 0:
                ratio = recent[i] / recent[i - 1]
                if 0 < ratio < 1:
                    ratios.append(ratio)
        return float(np.mean(ratios)) if ratios else 0.0

    def clear(self):
        for lst in (self.residual_norms, self.iteration_times, self.precision_levels, self.grid_levels):
            lst.clear()


class BaseSolver:                                                            # solvers/base.py:67-180
    def __init__(self, max_iterations=1000, tolerance=1e-8, verbose=False, name="BaseSolver"):
        self.max_iterations, self.tolerance = max_iterations, tolerance
        self.verbose, self.name = verbose, name
        self.history = ConvergenceHistory()
        self.converged = False
        self.final_residual = float("inf")
        self.iterations_performed = 0

    def check_convergence(self, residual_norm, iteration):
        converged = residual_norm < self.tolerance
        if not converged and iteration >= self.max_iterations:
            logger.warning(f"{self.name} reached max iterations ({self.max_iterations}): "
                           f"residual = {residual_norm:.2e}")
        return converged

    def get_convergence_info(self):
        return {
            "converged": self.converged,
            "iterations": self.iterations_performed,
            "final_residual": self.final_residual,
            "convergence_rate": self.history.get_convergence_rate(),
            "residual_history": self.history.residual_norms.copy(),
            "total_time": sum(self.history.iteration_times),
            "average_time_per_iteration": (float(np.mean(self.history.iteration_times))
                                           if self.history.iteration_times else 0.0),
            "precision_levels_used": list(set(self.history.precision_levels)),
        }

    def reset(self):
        self.history.clear()
        self.converged = False
        self.final_residual = float("inf")
        self.iterations_performed = 0


def _norm(grid, field):
    out = C.c_double(0.0)
    _lib.check(_lib.load().mg_op_norm(_lib.dtype_code(field.dtype), grid.nx, grid.ny, grid.hx, grid.hy,
                                      _lib.ptr(field), C.byref(out)))
    return out.value


class IterativeSolver(BaseSolver):                                           # solvers/base.py:183-290
    #: how the C library names this smoother (mg_smoother_t); subclasses set it
    kind = None

    def __init__(self, max_iterations=1000, tolerance=1e-8, relaxation_parameter=1.0, verbose=False,
                 name="IterativeSolver"):
        super().__init__(max_iterations, tolerance, verbose, name)
        self.omega = relaxation_parameter
        if not 0 < relaxation_parameter <= 2:
            logger.warning(f"Relaxation parameter {relaxation_parameter} may cause instability")

    def smooth(self, grid, operator, u, rhs, num_iterations=1):
        raise NotImplementedError

    def _prep(self, grid, u, rhs):
        u = _lib.as_c(u)
        rhs = np.ascontiguousarray(rhs, dtype=u.dtype)
        if u.shape != grid.shape or rhs.shape != grid.shape:
            raise ValueError(f"Field shape {u.shape} doesn't match grid shape {grid.shape}")
        return u, rhs

    def solve(self, grid, operator, rhs, initial_guess=None, precision_manager=None):
        self.reset()
        u = np.zeros_like(rhs) if initial_guess is None else np.array(initial_guess, copy=True)
        iteration, residual_norm = 0, float("inf")
        for iteration in range(1, self.max_iterations + 1):
            t0 = time.time()
            u = self.smooth(grid, operator, u, rhs, 1)
            residual_norm = _norm(grid, operator.residual(grid, u, rhs))
            level = precision_manager.current_precision.value if precision_manager else "unknown"
            self.history.record_iteration(residual_norm, time.time() - t0, level)
            if self.check_convergence(residual_norm, iteration):
                self.converged = True
                break
        self.iterations_performed = iteration
        self.final_residual = residual_norm
        return u, self.get_convergence_info()


class JacobiSmoother(IterativeSolver):                                       # solvers/smoothers.py:16-86
    kind = _lib.MG_JACOBI

    def __init__(self, max_iterations=1000, tolerance=1e-8, relaxation_parameter=2.0 / 3.0, verbose=False):
        super().__init__(max_iterations, tolerance, relaxation_parameter, verbose, "Jacobi")

    def smooth(self, grid, operator, u, rhs, num_iterations=1):
        u, rhs = self._prep(grid, u, rhs)
        out = np.empty_like(u)
        if hasattr(operator, "field"):                 # DiffusionOperator: variable-coefficient Jacobi
            a = operator.field(grid, u.dtype)
            _lib.check(_lib.load().mg_op_jacobi_var(_lib.dtype_code(u.dtype), grid.nx, grid.ny, grid.hx, grid.hy,
                                                    float(self.omega), int(num_iterations), _lib.ptr(a), _lib.ptr(u),
                                                    _lib.ptr(rhs), _lib.ptr(out)))
            return out
        if getattr(operator, "shift", 0.0):            # HelmholtzOperator: the shift joins the divisor
            _lib.check(_lib.load().mg_op_helmholtz(_lib.dtype_code(u.dtype), 1, grid.nx, grid.ny, grid.hx, grid.hy, -1.0,
                                                   float(operator.shift), float(self.omega), int(num_iterations),
                                                   _lib.ptr(u), _lib.ptr(rhs), _lib.ptr(out)))
            return out
        _lib.check(_lib.load().mg_op_jacobi(_lib.dtype_code(u.dtype), grid.nx, grid.ny, grid.hx, grid.hy,
                                            float(self.omega), int(num_iterations), _lib.ptr(u), _lib.ptr(rhs),
                                            _lib.ptr(out)))
        return out


class WeightedJacobiSmoother(JacobiSmoother):                                # solvers/smoothers.py:210-225
    def __init__(self, max_iterations=1000, tolerance=1e-8, verbose=False):
        super().__init__(max_iterations, tolerance, 4.0 / 5.0, verbose)
        self.name = "WeightedJacobi"


class EnhancedJacobiSolver(JacobiSmoother):                                  # solvers/iterative.py:18-108
    def __init__(self, max_iterations=1000, tolerance=1e-8, relaxation_parameter=2.0 / 3.0, verbose=False,
                 use_vectorized=True):
        super().__init__(max_iterations, tolerance, relaxation_parameter, verbose)
        self.name = "EnhancedJacobi"
        self.use_vectorized = use_vectorized


class GaussSeidelSmoother(IterativeSolver):                                  # solvers/smoothers.py:89-207
    def __init__(self, max_iterations=1000, tolerance=1e-8, relaxation_parameter=1.0, verbose=False,
                 red_black=False):
        super().__init__(max_iterations, tolerance, relaxation_parameter, verbose, "Gauss-Seidel")
        self.red_black = red_black

    @property
    def kind(self):
        return _lib.MG_RBGS if self.red_black else _lib.MG_LEXGS

    def smooth(self, grid, operator, u, rhs, num_iterations=1):
        u, rhs = self._prep(grid, u, rhs)
        out = np.empty_like(u)
        lib = _lib.load()
        if self.red_black and hasattr(operator, "field"):      # DiffusionOperator: variable-coefficient red-black GS
            a = operator.field(grid, u.dtype)
            _lib.check(lib.mg_op_rbgs_var(_lib.dtype_code(u.dtype), grid.nx, grid.ny, grid.hx, grid.hy, float(self.omega),
                                          int(num_iterations), _lib.ptr(a), _lib.ptr(u), _lib.ptr(rhs), _lib.ptr(out)))
        elif self.red_black and getattr(operator, "shift", 0.0):
            _lib.check(lib.mg_op_helmholtz(_lib.dtype_code(u.dtype), 2, grid.nx, grid.ny, grid.hx, grid.hy, -1.0,
                                           float(operator.shift), float(self.omega), int(num_iterations), _lib.ptr(u),
                                           _lib.ptr(rhs), _lib.ptr(out)))
        elif self.red_black:
            _lib.check(lib.mg_op_rbgs(_lib.dtype_code(u.dtype), grid.nx, grid.ny, grid.hx, grid.hy,
                                      float(self.omega), int(num_iterations), _lib.ptr(u), _lib.ptr(rhs),
                                      _lib.ptr(out)))
        else:
            if self.omega != 1.0:
                raise NotImplementedError("lexicographic SOR (omega != 1) is outside the accelerated hot path")
            if num_iterations < 1:
                return u.copy()
            # exactly num_iterations lexicographic sweeps: tolerance < 0 never stops early
            _lib.check(lib.mg_op_coarse_solve(_lib.dtype_code(u.dtype), grid.nx, grid.ny, grid.hx, grid.hy,
                                              float(getattr(operator, "coefficient", -1.0)), -1.0,
                                              int(num_iterations), _lib.ptr(u), _lib.ptr(rhs), _lib.ptr(out), None))
        return out
