"""Time-dependent heat equation  du/dt = alpha Laplace(u) + f(x, y, t)  on top of the multigrid engine
(SURVEY.md section 8f rank 4).  Mirrors applications/heat_equation.py: TimeSteppingScheme, BoundaryType,
BoundaryCondition, HeatEquationConfig, HeatEquationSolver (same constructor, method names, result-dict keys and the
same order of operations inside every step), create_gaussian_initial_condition, create_time_dependent_boundary.

What differs, on purpose:
  * Implicit steps solve  (-Laplace_h + lambda) u = rhs'  (lambda = 1/(alpha dt) or 2/(alpha dt),
    heat_equation.py:209-220, 254-261) with multigrid V-cycles of the SHIFTED operator on the GPU (mg_set_shift).
    The reference builds a CorrectedMultigridSolver (:101-106) but never calls it: its _solve_helmholtz (:459-497)
    runs lexicographic Gauss-Seidel sweeps of the same linear system, and because the residual it tests carries a sign
    error (:474, `rhs - lap + lambda u` for `rhs + lap - lambda u`) the stop test never fires and every step costs
    exactly 100 sweeps.  Both iterate on the same equations from the same initial guess (u_old, boundary ring
    included), so the reference's result is ours up to ITS remaining iteration error (tests/golden/heat.npz).
  * The rhs ring is cleared before the solve: the interior equations never read it, but the engine's norm counts
    r = f on the ring and would never pass the tolerance otherwise.
  * The Laplacian of explicit / Crank-Nicolson steps is evaluated by the HIP residual kernel (same values as
    _compute_laplacian :430-442 up to rounding: a different association of the same five terms).
  * Source terms and boundary data are evaluated on whole coordinate arrays when the callable accepts them (the
    reference loops over points); scalar-only callables fall back to np.vectorize.
"""
import logging
import time
from dataclasses import dataclass
from enum import Enum
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np

from . import _lib
from .engine import MultigridEngine
from .grid import Grid
from .operators import LaplacianOperator
from .precision import PrecisionManager

logger = logging.getLogger(__name__)


class TimeSteppingScheme(Enum):                                          # heat_equation.py:25-30
    EXPLICIT_EULER = "explicit_euler"
    IMPLICIT_EULER = "implicit_euler"
    CRANK_NICOLSON = "crank_nicolson"
    BDF2 = "bdf2"


class BoundaryType(Enum):                                                # heat_equation.py:33-38
    DIRICHLET = "dirichlet"
    NEUMANN = "neumann"
    ROBIN = "robin"
    PERIODIC = "periodic"


@dataclass
class BoundaryCondition:                                                 # heat_equation.py:41-53
    boundary_type: BoundaryType
    value: Optional[Callable[[float, float, float], float]] = None      # f(x, y, t)
    alpha: Optional[float] = None                                       # Robin: alpha u + beta du/dn = g
    beta: Optional[float] = None

    def evaluate(self, x, y, t):
        if self.value is None:
            return 0.0
        return self.value(x, y, t)


@dataclass
class HeatEquationConfig:                                                # heat_equation.py:56-72
    thermal_diffusivity: float = 1.0
    initial_condition: Callable[[float, float], float] = None
    source_term: Optional[Callable[[float, float, float], float]] = None
    boundary_conditions: Dict[str, BoundaryCondition] = None

    def __post_init__(self):
        if self.boundary_conditions is None:
            zero = lambda x, y, t: 0.0                                   # noqa: E731
            self.boundary_conditions = {k: BoundaryCondition(BoundaryType.DIRICHLET, zero)
                                        for k in ("left", "right", "bottom", "top")}


def _on_arrays(fn, *args):
    """fn evaluated on broadcast coordinate arrays; callables written for scalars go through np.vectorize."""
    shape = np.broadcast(*[np.asarray(a) for a in args]).shape
    try:
        out = np.asarray(fn(*args), dtype=np.float64)
        if out.shape == shape:
            return out
        if out.shape == ():
            return np.full(shape, float(out))
    except (TypeError, ValueError):        # written for scalars (math.sin(x), `if x < 0.5`, ...): evaluate point by point
        pass
    return np.asarray(np.vectorize(fn, otypes=[np.float64])(*args), dtype=np.float64).reshape(shape)


class HeatEquationSolver:
    """du/dt = alpha Laplace(u) + f  (heat_equation.py:75-600), implicit steps by shifted multigrid on the GPU."""

    def __init__(self, config: HeatEquationConfig, grid: Grid, precision_manager: Optional[PrecisionManager] = None,
                 device_id: int = 0, max_levels: int = 32, smoother: str = "jacobi"):
        self.config = config
        self.grid = grid
        self.precision_manager = precision_manager or PrecisionManager()
        # heat_equation.py:101-106: max_iterations = 20, tolerance = 1e-10 (there for a solver that is never called)
        self.mg_max_iterations, self.mg_tolerance = 20, 1e-10
        sm, omega = (_lib.MG_JACOBI, 0.8) if smoother == "jacobi" else (_lib.MG_RBGS, 1.0)
        self.mg_solver = MultigridEngine(grid.nx, grid.ny, tuple(float(v) for v in grid.domain), -1.0, max_levels, "V",
                                         2, 2, sm, omega, device=device_id)
        self.laplacian = LaplacianOperator(coefficient=1.0)
        self.current_time = 0.0
        self.current_solution = None
        self.solution_history, self.time_history, self.dt_history = [], [], []
        self.helmholtz_stats = []        # per implicit solve: (lambda, cycles, final ||r||)
        self._x = np.linspace(grid.domain[0], grid.domain[1], grid.nx)   # heat_equation.py:134-135, 449-450, 501-502
        self._y = np.linspace(grid.domain[2], grid.domain[3], grid.ny)
        logger.info(f"Initialized HeatEquationSolver: alpha={config.thermal_diffusivity}")

    # -- initial condition (heat_equation.py:120-153) -----------------------------------------
    def set_initial_condition(self, u0: Optional[np.ndarray] = None) -> np.ndarray:
        if u0 is not None:
            self.current_solution = np.array(u0, dtype=np.float64)
        elif self.config.initial_condition is not None:
            self.current_solution = _on_arrays(self.config.initial_condition, self._x[:, None], self._y[None, :]).copy()
        else:
            self.current_solution = np.zeros((self.grid.nx, self.grid.ny))
        self._apply_boundary_conditions(self.current_solution, 0.0)
        self.current_time = 0.0
        self.solution_history = [self.current_solution.copy()]
        self.time_history = [0.0]
        return self.current_solution

    # -- single steps ---------------------------------------------------------------------
    def explicit_euler_step(self, u_old, dt):                            # heat_equation.py:155-185
        h_min = min(self.grid.hx, self.grid.hy)
        dt_stable = h_min**2 / (4 * self.config.thermal_diffusivity)
        if dt > dt_stable:
            logger.warning(f"Time step dt={dt:.2e} exceeds stability limit {dt_stable:.2e}")
        lap_u = self._compute_laplacian(u_old)
        source = self._evaluate_source_term(self.current_time)
        u_new = u_old + dt * (self.config.thermal_diffusivity * lap_u + source)
        self._apply_boundary_conditions(u_new, self.current_time + dt)
        return u_new

    def implicit_euler_step(self, u_old, dt):                            # heat_equation.py:187-225
        alpha = self.config.thermal_diffusivity
        source = self._evaluate_source_term(self.current_time + dt)
        rhs = u_old + dt * source
        self._apply_boundary_conditions_to_rhs(rhs, self.current_time + dt)
        lambda_coeff = 1.0 / (dt * alpha)
        mg_rhs = rhs / (dt * alpha)
        u_new = self._solve_helmholtz(mg_rhs, lambda_coeff, u_old)
        self._apply_boundary_conditions(u_new, self.current_time + dt)
        return u_new

    def crank_nicolson_step(self, u_old, dt):                            # heat_equation.py:227-266
        alpha = self.config.thermal_diffusivity
        lap_u_old = self._compute_laplacian(u_old)
        source_old = self._evaluate_source_term(self.current_time)
        source_new = self._evaluate_source_term(self.current_time + dt)
        rhs = u_old + dt * alpha * lap_u_old / 2 + dt * (source_old + source_new) / 2
        self._apply_boundary_conditions_to_rhs(rhs, self.current_time + dt)
        lambda_coeff = 2.0 / (dt * alpha)
        mg_rhs = 2.0 * rhs / (dt * alpha)
        u_new = self._solve_helmholtz(mg_rhs, lambda_coeff, u_old)
        self._apply_boundary_conditions(u_new, self.current_time + dt)
        return u_new

    def adaptive_time_stepping(self, u_old, dt_initial, error_tolerance=1e-4,
                               scheme=TimeSteppingScheme.CRANK_NICOLSON) -> Tuple[np.ndarray, float]:
        """Step doubling (heat_equation.py:268-330): one step of dt against two of dt/2, Richardson estimate."""
        dt = dt_initial
        max_iterations = 10
        safety_factor = 0.8
        u_full = u_old
        for _ in range(max_iterations):
            u_full = self._single_time_step(u_old, dt, scheme)
            u_half1 = self._single_time_step(u_old, dt / 2, scheme)
            u_half2 = self._single_time_step(u_half1, dt / 2, scheme)
            if scheme in (TimeSteppingScheme.EXPLICIT_EULER, TimeSteppingScheme.IMPLICIT_EULER):
                error_est = np.linalg.norm(u_half2 - u_full)
                order = 1
            else:
                error_est = np.linalg.norm(u_half2 - u_full) / 3.0
                order = 2
            if error_est < error_tolerance:
                return u_half2, dt
            dt = max(dt / 4, dt * safety_factor * (error_tolerance / error_est) ** (1 / (order + 1)))
        logger.warning(f"Adaptive time stepping failed to converge after {max_iterations} iterations")
        return u_full, dt

    # -- time loop (heat_equation.py:332-417) -----------------------------------------------
    def solve_time_dependent(self, t_final, dt_initial=None, scheme=TimeSteppingScheme.CRANK_NICOLSON, adaptive=True,
                             error_tolerance=1e-4, save_interval=1) -> Dict[str, Any]:
        if self.current_solution is None:
            raise ValueError("Initial condition not set. Call set_initial_condition() first.")
        if dt_initial is None:
            h_min = min(self.grid.hx, self.grid.hy)
            if scheme == TimeSteppingScheme.EXPLICIT_EULER:
                dt_initial = 0.2 * h_min**2 / self.config.thermal_diffusivity
            else:
                dt_initial = 0.1 * h_min
        dt = dt_initial
        step_count = 0
        start_time = time.time()
        while self.current_time < t_final:
            if self.current_time + dt > t_final:
                dt = t_final - self.current_time
            if adaptive and scheme != TimeSteppingScheme.EXPLICIT_EULER:
                u_new, dt = self.adaptive_time_stepping(self.current_solution, dt, error_tolerance, scheme)
            else:
                u_new = self._single_time_step(self.current_solution, dt, scheme)
            self.current_solution = u_new
            self.current_time += dt
            step_count += 1
            if step_count % save_interval == 0:
                self.solution_history.append(u_new.copy())
                self.time_history.append(self.current_time)
                self.dt_history.append(dt)
        solve_time = time.time() - start_time
        return {"solution_history": self.solution_history, "time_history": self.time_history,
                "dt_history": self.dt_history, "final_solution": self.current_solution,
                "final_time": self.current_time, "total_steps": step_count, "solve_time": solve_time,
                "scheme": scheme.value, "adaptive": adaptive}

    def _single_time_step(self, u_old, dt, scheme):                      # heat_equation.py:419-428
        if scheme == TimeSteppingScheme.EXPLICIT_EULER:
            return self.explicit_euler_step(u_old, dt)
        if scheme == TimeSteppingScheme.IMPLICIT_EULER:
            return self.implicit_euler_step(u_old, dt)
        if scheme == TimeSteppingScheme.CRANK_NICOLSON:
            return self.crank_nicolson_step(u_old, dt)
        raise ValueError(f"Unsupported time stepping scheme: {scheme}")

    # -- pieces ---------------------------------------------------------------------------
    def _compute_laplacian(self, u):                                     # heat_equation.py:430-442; boundary 0
        return self.laplacian.apply(self.grid, np.ascontiguousarray(u, dtype=np.float64))

    def _evaluate_source_term(self, t):                                  # heat_equation.py:444-457
        if self.config.source_term is None:
            return np.zeros((self.grid.nx, self.grid.ny))
        return _on_arrays(self.config.source_term, self._x[:, None], self._y[None, :], t)

    def _solve_helmholtz(self, rhs, lambda_coeff, initial_guess):        # replaces heat_equation.py:459-497
        """(-Laplace_h + lambda) u = rhs on the interior, boundary ring of u = that of initial_guess."""
        f = np.array(rhs, dtype=np.float64)
        f[0, :] = f[-1, :] = 0.0
        f[:, 0] = f[:, -1] = 0.0
        eng = self.mg_solver
        eng.set_shift(lambda_coeff)
        scale = max(1.0, float(np.sqrt(self.grid.hx * self.grid.hy * np.sum(f * f))))
        u, info = eng.solve(f, np.ascontiguousarray(initial_guess, dtype=np.float64), self.mg_tolerance * scale,
                            self.mg_max_iterations)
        self.helmholtz_stats.append((lambda_coeff, info["iterations"],
                                     info["residual_history"][-1] if info["iterations"] else info["initial_residual"]))
        return u

    def _edge(self, location):
        x, y = self._x, self._y
        return {"left": (np.s_[0, :], np.s_[1, :], x[0], y), "right": (np.s_[-1, :], np.s_[-2, :], x[-1], y),
                "bottom": (np.s_[:, 0], np.s_[:, 1], x, y[0]), "top": (np.s_[:, -1], np.s_[:, -2], x, y[-1])}[location]

    def _apply_boundary_conditions(self, u, t):                          # heat_equation.py:499-577, same order
        for location in ("left", "right", "bottom", "top"):
            bc = self.config.boundary_conditions.get(location)
            if bc:
                self._apply_single_boundary(u, bc, location, self._x, self._y, t)

    def _apply_single_boundary(self, u, bc, location, x, y, t):
        edge, inner, ex, ey = self._edge(location)
        if bc.boundary_type == BoundaryType.DIRICHLET:
            u[edge] = _on_arrays(bc.evaluate, ex, ey, t)
        elif bc.boundary_type == BoundaryType.NEUMANN:                   # one-sided: u_b = u_inner -/+ h g (:548-562)
            h = min(self.grid.hx, self.grid.hy)
            sign = -1.0 if location in ("left", "bottom") else 1.0
            u[edge] = u[inner] + sign * h * _on_arrays(bc.evaluate, ex, ey, t)
        elif bc.boundary_type == BoundaryType.ROBIN:                     # only 'left' exists in the reference (:570-577)
            if location == "left":
                h = min(self.grid.hx, self.grid.hy)
                g = _on_arrays(bc.evaluate, ex, ey, t)
                u[edge] = (g + bc.beta * u[inner] / h) / (bc.alpha + bc.beta / h)

    def _apply_boundary_conditions_to_rhs(self, rhs, t):                 # heat_equation.py:579-599
        for name, bc in self.config.boundary_conditions.items():
            if bc.boundary_type == BoundaryType.DIRICHLET and name in ("left", "right", "bottom", "top"):
                edge, _, ex, ey = self._edge(name)
                rhs[edge] = _on_arrays(bc.evaluate, ex, ey, t)


def create_gaussian_initial_condition(center=(0.5, 0.5), width=0.1, amplitude=1.0):    # heat_equation.py:602-610
    def gaussian(x, y):
        dx = x - center[0]
        dy = y - center[1]
        return amplitude * np.exp(-(dx**2 + dy**2) / (2 * width**2))
    return gaussian


def create_time_dependent_boundary(amplitude=1.0, frequency=1.0):                      # heat_equation.py:613-618
    def time_varying(x, y, t):
        return amplitude * np.sin(2 * np.pi * frequency * t)
    return time_varying
