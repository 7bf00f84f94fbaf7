"""The README-level API of the reference (README.md:73-93), which its code never defined
(SURVEY.md F1): multigrid.solvers.MixedPrecisionMultigrid and multigrid.problems.PoissonProblem.

    problem = PoissonProblem(source_term, nx=129, ny=129)
    solver = MixedPrecisionMultigrid(precision_strategy='adaptive', switch_threshold=1e-6, use_gpu=True)
    solution, info = solver.solve(problem)      # info['iterations'], info['residual'], info['solve_time']

use_gpu=True runs the MI355X engine.  use_gpu=False selects the REFERENCE's own CPU MultigridSolver
(it is not re-implemented here): it is loaded from MG_REFERENCE_SRC (the reference's src/ directory)
and raises if that is not available -- there is no silent CPU fallback for the GPU path.
"""
import importlib.util
import os
import sys
import time

import numpy as np

from . import _lib
from .grid import Grid
from .operators import LaplacianOperator, ProlongationOperator, RestrictionOperator
from .precision import PrecisionManager
from .smoothers import GaussSeidelSmoother, JacobiSmoother
from .solver import MultigridSolver


class PoissonProblem:
    """-Laplace(u) = f on a rectangle, Dirichlet data on the boundary (homogeneous by default).

    source_term(x, y) is evaluated on the 'ij' mesh like the reference's Grid.X/Grid.Y
    (core/grid.py:48-50; applications/poisson_solver.py:24-32 for the optional fields)."""

    def __init__(self, source_term, nx=129, ny=129, domain=(0.0, 1.0, 0.0, 1.0), boundary_values=None,
                 analytical_solution=None, name="poisson"):
        if nx < 3 or ny < 3:
            raise ValueError("Grid must have at least 3 points in each direction")
        self.source_term = source_term
        self.nx, self.ny = int(nx), int(ny)
        self.domain = tuple(domain)
        self.boundary_values = boundary_values
        self.analytical_solution = analytical_solution
        self.name = name

    def grid(self, dtype=np.float64):
        return Grid(self.nx, self.ny, self.domain, dtype)

    def rhs(self, dtype=np.float64):
        """f on the mesh, built one row-block at a time (no n^2 meshgrid pair is kept)."""
        x = np.linspace(self.domain[0], self.domain[1], self.nx, dtype=np.float64)
        y = np.linspace(self.domain[2], self.domain[3], self.ny, dtype=np.float64)
        out = np.empty((self.nx, self.ny), dtype=dtype)
        step = max(1, (1 << 22) // max(self.ny, 1))
        for a in range(0, self.nx, step):
            X, Y = np.meshgrid(x[a:a + step], y, indexing="ij")
            out[a:a + step] = np.asarray(self.source_term(X, Y), dtype=np.float64)
        return out

    def initial_guess(self, dtype=np.float64):
        """Zero interior; Dirichlet data, if any, on the boundary ring."""
        if self.boundary_values is None:
            return None
        u0 = np.zeros((self.nx, self.ny), dtype=dtype)
        x = np.linspace(self.domain[0], self.domain[1], self.nx)
        y = np.linspace(self.domain[2], self.domain[3], self.ny)
        bv = self.boundary_values
        if callable(bv):
            u0[0, :], u0[-1, :] = bv(x[0], y), bv(x[-1], y)
            u0[:, 0], u0[:, -1] = bv(x, y[0]), bv(x, y[-1])
        else:
            u0[0, :] = u0[-1, :] = u0[:, 0] = u0[:, -1] = float(bv)
        return u0


def default_max_levels(nx, ny):
    """Deepest hierarchy the reference's rules allow: coarsen while (n-1) is even and the
    coarse grid keeps n >= 5 (solvers/multigrid.py:153-171)."""
    levels = 1
    while (nx - 1) % 2 == 0 and (ny - 1) % 2 == 0 and (nx - 1) // 2 + 1 >= 5 and (ny - 1) // 2 + 1 >= 5:
        nx, ny = (nx - 1) // 2 + 1, (ny - 1) // 2 + 1
        levels += 1
    return levels


def _load_reference():
    src = os.environ.get("MG_REFERENCE_SRC")
    if not src or not os.path.isdir(os.path.join(src, "multigrid")):
        raise RuntimeError(
            "use_gpu=False selects the reference's own CPU solver, which is not part of this package: "
            "set MG_REFERENCE_SRC to the reference's src/ directory (the GPU path has no CPU fallback)")
    name = "_mg_reference_cpu"
    if name in sys.modules:
        return sys.modules[name]
    pkg = os.path.join(src, "multigrid")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg, "__init__.py"),
                                                  submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    return mod


class MixedPrecisionMultigrid:
    STRATEGIES = ("double", "single", "mixed", "adaptive", "adaptive_reference", "defect")

    def __init__(self, precision_strategy="adaptive", switch_threshold=1e-6, use_gpu=True, max_levels=None,
                 max_iterations=50, tolerance=1e-8, cycle_type="V", pre_smooth_iterations=2,
                 post_smooth_iterations=2, smoother="jacobi", relaxation_parameter=None, device_id=0,
                 coarse_tolerance=1e-12, coarse_max_iterations=1000, use_fmg=False, fmg_cycles=1, n_gpus=1, device_ids=None,
                 decomposition_strategy="block", agglomerate_at=1025):
        """n_gpus > 1 (or device_ids with more than one entry, or a torch.distributed job of more than one rank): the solve
        runs block domain-decomposed through DistributedMultigridSolver (multi_gpu.py) -- one process per GPU under
        torch.distributed, virtual ranks on one GPU otherwise -- and returns the single-GPU solve's iterate."""
        if precision_strategy not in self.STRATEGIES:
            raise ValueError(f"Unknown precision strategy: {precision_strategy}")
        if smoother not in ("jacobi", "gauss_seidel", "red_black", "sor"):
            raise ValueError(f"Unknown smoother: {smoother}")
        self.precision_strategy = precision_strategy
        self.switch_threshold = switch_threshold
        self.use_gpu = use_gpu
        self.max_levels = max_levels
        self.max_iterations, self.tolerance = max_iterations, tolerance
        self.cycle_type = cycle_type
        self.pre, self.post = pre_smooth_iterations, post_smooth_iterations
        self.smoother = smoother
        self.omega = relaxation_parameter
        self.device_id = device_id
        self.coarse_tolerance, self.coarse_max_iterations = coarse_tolerance, coarse_max_iterations
        self.use_fmg, self.fmg_cycles = use_fmg, fmg_cycles
        self.device_ids = list(device_ids) if device_ids is not None else ([device_id] * int(n_gpus) if int(n_gpus) > 1 else None)
        self.n_gpus = len(self.device_ids) if self.device_ids else 1
        self.decomposition_strategy, self.agglomerate_at = decomposition_strategy, agglomerate_at
        if use_gpu:
            _lib.load()                                 # fail at construction, not at first solve
            if _lib.device_count() <= device_id:
                raise RuntimeError(f"mghip: HIP device {device_id} is not available (use_gpu=True has no CPU fallback)")

    # ---------------------------------------------------------------------------------------
    def _precision_manager(self, mod=None):
        cls = PrecisionManager if mod is None else mod.core.precision.PrecisionManager
        s = self.precision_strategy
        if s == "double":
            return None
        if s == "single":
            return None                                  # Grid(dtype=float32)
        if s == "mixed":
            return cls(default_precision="mixed", convergence_threshold=self.switch_threshold)
        if s == "defect":          # defect correction: fp64 iterate / residual, fp32 cycles on the error equation (ours, GPU only)
            pm = cls(default_precision="double", adaptive=False, convergence_threshold=self.switch_threshold)
            pm.defect_correction = True
            return pm
        pm = cls(default_precision="double", adaptive=True, convergence_threshold=self.switch_threshold)
        pm.reference_rule = (s == "adaptive_reference")
        return pm

    def _smoother(self, mod=None):
        sm = (mod.solvers.smoothers if mod is not None else sys.modules[__package__ + ".smoothers"])
        if self.smoother == "jacobi":
            if mod is not None:                          # NumPy-vectorised twin (iterative.py:72-108)
                return mod.solvers.iterative.EnhancedJacobiSolver(relaxation_parameter=self.omega or 0.8)
            return JacobiSmoother(relaxation_parameter=self.omega or 0.8)
        omega = self.omega or (1.15 if self.smoother == "sor" else 1.0)
        return sm.GaussSeidelSmoother(relaxation_parameter=omega, red_black=True)

    def solve(self, problem, initial_guess=None):
        dtype = np.float32 if self.precision_strategy == "single" else np.float64
        levels = self.max_levels or default_max_levels(problem.nx, problem.ny)
        rhs = problem.rhs(dtype)
        u0 = initial_guess if initial_guess is not None else problem.initial_guess(dtype)
        t0 = time.time()
        from .multi_gpu import DistributedMultigridSolver, _dist_if_initialised
        if self.use_gpu and (self.n_gpus > 1 or _dist_if_initialised() is not None):
            if self.precision_strategy in ("adaptive_reference", "defect") or self.use_fmg:
                raise NotImplementedError("the decomposed solver runs the 'double', 'single', 'mixed' and 'adaptive' strategies "
                                          "without a full-multigrid start")
            grid = problem.grid(dtype)
            name = {"jacobi": "jacobi", "gauss_seidel": "gauss_seidel", "red_black": "gauss_seidel", "sor": "sor"}[self.smoother]
            solver = DistributedMultigridSolver(device_ids=self.device_ids, decomposition_strategy=self.decomposition_strategy,
                                                agglomerate_at=self.agglomerate_at, max_levels=levels,
                                                max_iterations=self.max_iterations, tolerance=self.tolerance,
                                                cycle_type=self.cycle_type, pre_smooth_iterations=self.pre,
                                                post_smooth_iterations=self.post, smoother=name, relaxation_parameter=self.omega,
                                                coarse_tolerance=self.coarse_tolerance,
                                                coarse_max_iterations=self.coarse_max_iterations)
            op = LaplacianOperator(coefficient=-1.0)
            solver.setup(grid, op, RestrictionOperator("full_weighting"), ProlongationOperator("bilinear"))
            try:
                u, info = solver.solve(grid, op, rhs, u0, self._precision_manager())
            finally:
                solver.cleanup()
        elif self.use_gpu:
            grid = problem.grid(dtype)
            solver = MultigridSolver(levels, self.max_iterations, self.tolerance, self.cycle_type, self.pre, self.post,
                                     self.coarse_tolerance, self.coarse_max_iterations, device_id=self.device_id,
                                     fmg_cycles=self.fmg_cycles if self.use_fmg else 0)
            solver.setup(grid, LaplacianOperator(coefficient=-1.0), RestrictionOperator("full_weighting"),
                         ProlongationOperator("bilinear"), smoother=self._smoother())
            try:
                u, info = solver.solve(grid, solver.operators[0], rhs, u0, self._precision_manager())
            finally:
                solver.cleanup()
        else:
            if self.precision_strategy == "defect":
                raise NotImplementedError("defect correction is a device policy (mg_config.precision = MG_PREC_DEFECT); use_gpu=True")
            mod = _load_reference()
            grid = mod.core.grid.Grid(problem.nx, problem.ny, problem.domain, dtype)
            op = mod.operators.laplacian.LaplacianOperator(coefficient=-1.0)
            solver = mod.solvers.multigrid.MultigridSolver(
                max_levels=levels, max_iterations=self.max_iterations, tolerance=self.tolerance,
                cycle_type=self.cycle_type, pre_smooth_iterations=self.pre, post_smooth_iterations=self.post,
                coarse_tolerance=self.coarse_tolerance, coarse_max_iterations=self.coarse_max_iterations)
            solver.setup(grid, op, mod.operators.transfer.RestrictionOperator("full_weighting"),
                         mod.operators.transfer.ProlongationOperator("bilinear"), smoother=self._smoother(mod))
            pm = self._precision_manager(mod)
            if pm is not None and self.precision_strategy == "adaptive":
                raise NotImplementedError("the reference CPU solver has no working one-way adaptive rule "
                                          "(SURVEY.md F11); use 'adaptive_reference', 'mixed', 'double' or 'single'")
            u, info = solver.solve(grid, op, rhs, u0, pm)
        info = dict(info)
        info["residual"] = info["final_residual"]        # README.md:90-92 keys
        info["solve_time"] = time.time() - t0
        info["precision_strategy"] = self.precision_strategy
        info["use_gpu"] = self.use_gpu
        if problem.analytical_solution is not None:
            x = np.linspace(problem.domain[0], problem.domain[1], problem.nx)
            y = np.linspace(problem.domain[2], problem.domain[3], problem.ny)
            X, Y = np.meshgrid(x, y, indexing="ij")
            info["max_error"] = float(np.max(np.abs(u - problem.analytical_solution(X, Y))))
        return u, info
