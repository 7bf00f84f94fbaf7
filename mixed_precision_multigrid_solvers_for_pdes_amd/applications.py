"""Callers of the hot path, mirrored from the reference so that its
end-to-end examples run on the MI355X engine:

  * PoissonProblem (dataclass) + PoissonSolver2D -- applications/poisson_solver.py:24-32, 35-420: same constructor
    kwargs, same result-dict keys, same error norms and convergence-study arithmetic.  Differences, on purpose:
    the operator is the self-consistent LaplacianOperator(coefficient=-1.0) (the reference builds the default +1
    operator and diverges); Dirichlet data (constant or callable) goes into the boundary ring of the
    initial guess (the reference writes to a non-existent `grid.data`, poisson_solver.py:203-207); Neumann / mixed
    conditions, which the reference only sketches by patching the rhs, are rejected.
  * MultigridPreconditioner -- preconditioning/multigrid_preconditioner.py:20-176: a fixed number of cycles from a
    zero guess on the device-resident hierarchy (no convergence test), for Krylov outer loops.
"""
import logging
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from . import _lib
from .engine import MultigridEngine
from .grid import Grid
from .operators import LaplacianOperator, ProlongationOperator, RestrictionOperator
from .smoothers import GaussSeidelSmoother, JacobiSmoother
from .solver import GPUCommunicationAvoidingMultigrid, GPUMultigridSolver, MultigridSolver

logger = logging.getLogger(__name__)


@dataclass
class PoissonProblem:                                                    # applications/poisson_solver.py:24-32
    name: str
    source_function: Callable[[np.ndarray, np.ndarray], np.ndarray]
    analytical_solution: Optional[Callable[[np.ndarray, np.ndarray], np.ndarray]] = None
    boundary_conditions: Optional[Dict[str, Any]] = None
    domain: Tuple[float, float, float, float] = (0, 1, 0, 1)
    description: str = ""


class PoissonSolver2D:
    def __init__(self, solver_type="multigrid", max_levels=6, max_iterations=100, tolerance=1e-8, cycle_type="V",
                 use_gpu=True, device_id=0, enable_mixed_precision=False, smoother="jacobi"):
        if solver_type not in ("multigrid", "gpu_multigrid", "gpu_ca_multigrid"):
            raise ValueError(f"Unknown solver type: {solver_type}")
        if not use_gpu:
            raise RuntimeError("PoissonSolver2D(use_gpu=False) is the reference's own CPU path; this package only "
                               "provides the MI355X engine (no CPU fallback)")
        self.solver_type, self.max_levels = solver_type, max_levels
        self.max_iterations, self.tolerance = max_iterations, tolerance
        self.cycle_type, self.use_gpu, self.device_id = cycle_type, use_gpu, device_id
        self.enable_mixed_precision = enable_mixed_precision
        self.solver = self._create_solver(smoother)
        self.operator = LaplacianOperator(coefficient=-1.0)
        self.restriction = RestrictionOperator("full_weighting")
        self.prolongation = ProlongationOperator("bilinear")
        self.current_problem = None
        self.solve_history: List[Dict[str, Any]] = []

    def _create_solver(self, smoother="jacobi"):                                          # poisson_solver.py:88-116
        kw = dict(device_id=self.device_id, max_levels=self.max_levels, max_iterations=self.max_iterations,
                  tolerance=self.tolerance, cycle_type=self.cycle_type, smoother=smoother,
                  enable_mixed_precision=self.enable_mixed_precision)
        if self.solver_type == "gpu_ca_multigrid":
            return GPUCommunicationAvoidingMultigrid(use_fmg=True, **kw)
        return GPUMultigridSolver(**kw)

    def _boundary_guess(self, grid, bc):
        """Dirichlet data -> boundary ring of the initial guess (None for homogeneous data)."""
        if not bc:
            return None
        kind = bc.get("type", "dirichlet")
        if kind != "dirichlet":
            raise NotImplementedError(f"boundary condition type {kind!r}: the multigrid path solves Dirichlet problems")
        value = bc.get("value", 0.0)
        u0 = np.zeros(grid.shape)
        if callable(value):
            u0[0, :], u0[-1, :] = value(grid.x[0], grid.y), value(grid.x[-1], grid.y)
            u0[:, 0], u0[:, -1] = value(grid.x, grid.y[0]), value(grid.x, grid.y[-1])
        elif value == 0.0:
            return None
        else:
            u0[0, :] = u0[-1, :] = u0[:, 0] = u0[:, -1] = float(value)
        return u0

    def solve_poisson_problem(self, problem, nx, ny, initial_guess=None):                # poisson_solver.py:118-189
        self.current_problem = problem
        grid = Grid(nx=nx, ny=ny, domain=tuple(float(v) for v in problem.domain))
        rhs = np.asarray(problem.source_function(grid.X, grid.Y), dtype=np.float64)
        ring = self._boundary_guess(grid, problem.boundary_conditions)
        if initial_guess is None:
            initial_guess = ring
        elif ring is not None:
            initial_guess = np.array(initial_guess, dtype=np.float64, copy=True)
            initial_guess[0, :], initial_guess[-1, :] = ring[0, :], ring[-1, :]
            initial_guess[:, 0], initial_guess[:, -1] = ring[:, 0], ring[:, -1]
        self.solver.setup(grid, self.operator, self.restriction, self.prolongation)
        t0 = time.time()
        solution, solve_info = self.solver.solve(grid, self.operator, rhs, initial_guess)
        solve_time = time.time() - t0
        errors = {}
        if problem.analytical_solution:
            errors = self._compute_errors(solution, problem.analytical_solution(grid.X, grid.Y), grid)
        results = {"problem_name": problem.name, "grid_size": (nx, ny), "domain": problem.domain, "solution": solution,
                   "solve_time": solve_time, "solver_info": solve_info, "errors": errors,
                   "solver_type": self.solver_type, "use_gpu": self.use_gpu,
                   "mixed_precision": self.enable_mixed_precision}
        if problem.analytical_solution:
            results["analytical_solution"] = problem.analytical_solution(grid.X, grid.Y)
        self.solve_history.append(results)
        return results

    def _compute_errors(self, numerical, analytical, grid):                               # poisson_solver.py:281-313
        error = numerical - analytical
        l2_error = np.sqrt(np.sum(error**2) * grid.hx * grid.hy)
        l2_norm = np.sqrt(np.sum(analytical**2) * grid.hx * grid.hy)
        max_error = np.max(np.abs(error))
        max_norm = np.max(np.abs(analytical))
        gx = np.diff(error, axis=0) / grid.hx
        gy = np.diff(error, axis=1) / grid.hy
        h1 = np.sqrt(np.sum(gx[:-1, :]**2) * grid.hx * grid.hy + np.sum(gy[:, :-1]**2) * grid.hx * grid.hy)
        return {"l2_error": l2_error, "relative_l2_error": l2_error / l2_norm if l2_norm > 0 else l2_error,
                "max_error": max_error, "relative_max_error": max_error / max_norm if max_norm > 0 else max_error,
                "h1_semi_error": h1, "grid_spacing": (grid.hx, grid.hy)}

    def run_convergence_study(self, problem, grid_sizes, expected_order=2.0):             # poisson_solver.py:315-395
        if not problem.analytical_solution:
            raise ValueError("Convergence study requires analytical solution")
        results = [self.solve_poisson_problem(problem, nx, ny) for nx, ny in grid_sizes]
        rates = []
        for prev, cur in zip(results[:-1], results[1:]):
            h_ratio = min(prev["errors"]["grid_spacing"]) / min(cur["errors"]["grid_spacing"])
            l2_ratio = prev["errors"]["l2_error"] / cur["errors"]["l2_error"]
            max_ratio = prev["errors"]["max_error"] / cur["errors"]["max_error"]
            rates.append({"grid_transition": f"{prev['grid_size']} -> {cur['grid_size']}", "h_ratio": h_ratio,
                          "l2": np.log(l2_ratio) / np.log(h_ratio) if l2_ratio > 0 and h_ratio > 1 else 0,
                          "max": np.log(max_ratio) / np.log(h_ratio) if max_ratio > 0 and h_ratio > 1 else 0,
                          "l2_error_ratio": l2_ratio, "max_error_ratio": max_ratio})
        return {"problem_name": problem.name, "grid_sizes": grid_sizes, "results": results, "convergence_rates": rates,
                "expected_order": expected_order,
                "achieved_order": {"l2": np.mean([r["l2"] for r in rates]) if rates else 0,
                                   "max": np.mean([r["max"] for r in rates]) if rates else 0}}


class MultigridPreconditioner:                                         # preconditioning/multigrid_preconditioner.py:20-176
    def __init__(self, max_levels=3, cycle_type="V", pre_smooth_iterations=1, post_smooth_iterations=1, num_cycles=1,
                 coarse_tolerance=1e-6, coarse_max_iterations=100):
        self.name = "MultigridPreconditioner"
        self.max_levels, self.cycle_type = max_levels, cycle_type
        self.pre_smooth_iterations, self.post_smooth_iterations = pre_smooth_iterations, post_smooth_iterations
        self.num_cycles = num_cycles
        self.coarse_tolerance, self.coarse_max_iterations = coarse_tolerance, coarse_max_iterations
        self.grid = self.operator = self._engine = None
        self.setup_completed = False

    def setup(self, grid, operator, restriction_op=None, prolongation_op=None, smoother=None, coarse_solver=None,
              precision_manager=None):
        if restriction_op is not None and restriction_op.method != "full_weighting":
            raise NotImplementedError("the accelerated path implements full_weighting restriction")
        if prolongation_op is not None and prolongation_op.method != "bilinear":
            raise NotImplementedError("the accelerated path implements bilinear prolongation")
        if smoother is None:       # the reference defaults to lexicographic GS; its parallel twin is red-black GS
            smoother = GaussSeidelSmoother(red_black=True)
        self.grid, self.operator = grid, operator
        prec = _lib.MG_PREC_SINGLE if np.dtype(grid.dtype) == np.float32 else _lib.MG_PREC_DOUBLE
        if self._engine is not None:
            self._engine.close()
        self._engine = MultigridEngine(grid.nx, grid.ny, grid.domain, float(getattr(operator, "coefficient", -1.0)),
                                       self.max_levels, self.cycle_type, self.pre_smooth_iterations,
                                       self.post_smooth_iterations, smoother.kind, smoother.omega, self.coarse_tolerance,
                                       self.coarse_max_iterations, prec)
        self.setup_completed = True

    def apply(self, x):
        """z ~ A^{-1} x: num_cycles cycles from a zero guess, no convergence test (tolerance 1e-16 in the reference)."""
        if not self.setup_completed:
            raise RuntimeError("Multigrid preconditioner not setup")
        x = _lib.as_c(x)
        e = self._engine
        e.set_rhs(x)
        e.set_solution(None, x.dtype)
        e.cycle(self.num_cycles)
        return e.get_solution(x.dtype)

    def apply_transpose(self, x):
        return self.apply(x)

    def cleanup(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        self.setup_completed = False
