"""Solver drivers.  Mirror multigrid.solvers.multigrid.MultigridSolver (solvers/multigrid.py:28-391)
and multigrid.gpu.gpu_solver.GPUMultigridSolver (gpu/gpu_solver.py:24-501): same constructor kwargs,
setup()/solve() signatures, info-dict keys and errors.  The cycle runs device-resident in
libmghip.so; Python only translates the plugin objects into an mg_config."""
import logging
import time
import warnings

import numpy as np

from . import _lib
from .engine import MultigridEngine
from .precision import PrecisionLevel, PrecisionManager
from .smoothers import BaseSolver, GaussSeidelSmoother, IterativeSolver

logger = logging.getLogger(__name__)


class MultigridCycle:
    V_CYCLE = "V"
    W_CYCLE = "W"
    F_CYCLE = "F"


def _precision_config(grid, pm):
    """(mg precision policy, threshold, memory GB, reference_rule) for a Grid + PrecisionManager pair."""
    if np.dtype(grid.dtype) == np.float32:
        return _lib.MG_PREC_SINGLE, 1e-6, 4.0, False
    if pm is None:
        return _lib.MG_PREC_DOUBLE, 1e-6, 4.0, False
    thr, mem = pm.convergence_threshold, pm.memory_threshold_gb
    if getattr(pm, "defect_correction", False):
        # fp64 iterate and residual, fp32 cycles on the error equation (mg_config.precision = MG_PREC_DEFECT)
        return _lib.MG_PREC_DEFECT, thr, mem, False
    if pm.current_precision == PrecisionLevel.MIXED:
        # non-adaptive 'mixed' resolves to get_dtype(MIXED) = float64 on every level (precision.py:348-349)
        return (_lib.MG_PREC_MIXED_LEVELS if pm.adaptive else _lib.MG_PREC_DOUBLE), thr, mem, False
    if not pm.adaptive:
        return (_lib.MG_PREC_SINGLE_MANAGED if pm.current_precision == PrecisionLevel.SINGLE else _lib.MG_PREC_DOUBLE), thr, mem, False
    return _lib.MG_PREC_ADAPTIVE, thr, mem, bool(getattr(pm, "reference_rule", True))


class MultigridSolver(BaseSolver):
    def __init__(self, max_levels=4, max_iterations=50, tolerance=1e-8, cycle_type=MultigridCycle.V_CYCLE,
                 pre_smooth_iterations=2, post_smooth_iterations=2, coarse_tolerance=1e-12,
                 coarse_max_iterations=1000, verbose=False, device_id=0, profile=False, fmg_cycles=0, coarse_direct=None):
        """coarse_direct (ours): how a 5 x 5 coarsest grid is solved.  False: the reference's lexicographic Gauss-Seidel
        iteration to `coarse_tolerance` (solvers/multigrid.py:119-124, 355-370), bit for bit.  True: its nine unknowns
        directly.  None (default): directly, unless the environment variable MG_COARSE_DIRECT=0 asks for the iteration
        (W- / F-cycles visit the coarsest grid 2^(L-1) times per cycle and spent most of their time in it).  A direct solve satisfies
        coarse_tolerance exactly; residual histories then agree with the reference's to max(1e-9 relative,
        coarse_tolerance absolute) -- the accuracy the reference's own coarse solver is configured for -- and iterates to
        1e-12 relative (tests/test_gpu_solver.py)."""
        super().__init__(max_iterations, tolerance, verbose, "Multigrid")
        self.coarse_direct = coarse_direct
        self.max_levels = max_levels
        self.cycle_type = cycle_type
        self.pre_smooth_iterations = pre_smooth_iterations
        self.post_smooth_iterations = post_smooth_iterations
        self.coarse_tolerance = coarse_tolerance
        self.coarse_max_iterations = coarse_max_iterations
        self.device_id = device_id
        self.profile = profile
        self.fmg_cycles = fmg_cycles      # > 0: start from a full-multigrid guess when no initial guess is given
        self.grids = []
        self.operators, self.restriction_ops, self.prolongation_ops = [], [], []
        self.smoother = None
        self.coarse_solver = None
        self.level_stats = {}
        self._engines = {}
        self._setup_args = None

    # -- setup (solvers/multigrid.py:91-182) -------------------------------------------------
    def setup(self, fine_grid, operator, restriction_op, prolongation_op, smoother=None, coarse_solver=None):
        if smoother is None:
            # the reference's default (solvers/multigrid.py:112-117): a *lexicographic* Gauss-Seidel smoother.  It is
            # sequential by nature -- one workgroup sweeping anti-diagonals (MG_LEXGS) -- so the drop-in default is
            # parity, not speed: pass GaussSeidelSmoother(red_black=True) or a Jacobi smoother for the fast legs.
            smoother = GaussSeidelSmoother(max_iterations=max(self.pre_smooth_iterations, self.post_smooth_iterations),
                                           tolerance=self.tolerance * 0.1)
            if fine_grid.nx * fine_grid.ny > 513 * 513:
                warnings.warn("MultigridSolver.setup(smoother=None) keeps the reference's default lexicographic "
                              "Gauss-Seidel smoother, which runs on one workgroup; pass "
                              "GaussSeidelSmoother(red_black=True) or a Jacobi smoother for the bandwidth-bound legs")
        if not isinstance(smoother, IterativeSolver) or smoother.kind is None:
            raise TypeError("smoother must be a JacobiSmoother / GaussSeidelSmoother (or subclass)")
        if coarse_solver is not None and not (isinstance(coarse_solver, GaussSeidelSmoother) and not coarse_solver.red_black):
            raise NotImplementedError("the coarsest grid is solved by lexicographic Gauss-Seidel (the reference default)")
        if getattr(restriction_op, "method", None) != "full_weighting" or getattr(prolongation_op, "method", None) != "bilinear":
            raise NotImplementedError("the accelerated path implements full_weighting restriction and bilinear prolongation")
        if smoother.kind == _lib.MG_LEXGS and smoother.omega != 1.0:
            raise NotImplementedError("lexicographic SOR (omega != 1) is outside the accelerated hot path")
        coeff = float(getattr(operator, "coefficient", -1.0))
        if coeff > 0:
            warnings.warn("LaplacianOperator(coefficient > 0) is inconsistent with the smoothers, which relax "
                          "-Laplace(u) = rhs; the reference diverges in this configuration (use coefficient=-1.0)")
        self.smoother = smoother
        if coarse_solver is not None:
            self.coarse_tolerance = coarse_solver.tolerance
            self.coarse_max_iterations = coarse_solver.max_iterations
        self.coarse_solver = coarse_solver or GaussSeidelSmoother(max_iterations=self.coarse_max_iterations,
                                                                  tolerance=self.coarse_tolerance)
        self._build_hierarchy(fine_grid, operator, restriction_op, prolongation_op)
        self._setup_args = (fine_grid, coeff)
        self._coefficient_field = operator.field(fine_grid) if hasattr(operator, "field") else None
        for e in self._engines.values():
            e.close()
        self._engines = {}
        # build the default (no precision manager) hierarchy now so that setup fails early, like the reference
        self._engine(fine_grid, None)

    def _build_hierarchy(self, fine_grid, operator, restriction_op, prolongation_op):   # multigrid.py:135-182
        self.grids = [fine_grid]
        self.operators = [operator]
        self.restriction_ops, self.prolongation_ops = [], []
        current = fine_grid
        for _ in range(1, self.max_levels):
            try:
                coarse = current.coarsen()
            except ValueError:
                break
            if coarse.nx < 5 or coarse.ny < 5:
                break
            self.grids.append(coarse)
            self.operators.append(operator)
            self.restriction_ops.append(restriction_op)
            self.prolongation_ops.append(prolongation_op)
            current = coarse
        self.level_stats = {l: {"smooth_time": 0.0, "restrict_time": 0.0, "prolong_time": 0.0}
                            for l in range(len(self.grids))}

    def _engine(self, grid, pm):
        key = _precision_config(grid, pm)
        if key not in self._engines:
            prec, thr, mem, ref_rule = key
            g = self.grids[0]
            self._engines[key] = MultigridEngine(
                g.nx, g.ny, g.domain, self._setup_args[1], self.max_levels, self.cycle_type,
                self.pre_smooth_iterations, self.post_smooth_iterations, self.smoother.kind, self.smoother.omega,
                self.coarse_tolerance, self.coarse_max_iterations, prec, thr, mem, ref_rule,
                self.device_id, self.profile, fmg_cycles=self.fmg_cycles, coarse_direct=self.coarse_direct)
            if self._coefficient_field is not None:
                self._engines[key].set_coefficient(self._coefficient_field)
        return self._engines[key]

    # -- solve (solvers/multigrid.py:184-251) -------------------------------------------------
    def solve(self, grid, operator, rhs, initial_guess=None, precision_manager=None):
        if not self.grids or grid.shape != self.grids[0].shape:
            raise ValueError("Multigrid not properly setup or grid mismatch")
        self.reset()
        pm = precision_manager
        eng = self._engine(grid, pm)
        eng.set_shift(getattr(operator, "shift", 0.0))     # HelmholtzOperator: may change between solves (time steppers)
        rhs = np.asarray(rhs)
        work_dtype = np.float32 if np.dtype(grid.dtype) == np.float32 and rhs.dtype == np.float32 else np.float64
        t0 = time.time()
        u, r = eng.solve(rhs, initial_guess, self.tolerance, self.max_iterations, out_dtype=work_dtype)
        wall = time.time() - t0
        n = r["iterations"]
        names = {0: "float32", 1: "float64", 2: "mixed", 3: "defect"}
        per_it = r["solve_seconds"] / max(n, 1)
        for k in range(n):
            level = names[r["precision_codes"][k]] if pm is not None else "double"
            self.history.record_iteration(r["residual_history"][k], per_it, level, 0)
        self.converged = r["converged"]
        self.iterations_performed = n
        self.final_residual = r["residual_history"][-1] if n else float("inf")
        if not self.converged:
            logger.warning(f"{self.name} reached max iterations ({self.max_iterations}): "
                           f"residual = {self.final_residual:.2e}")
        if pm is not None and pm.adaptive and pm.current_precision != PrecisionLevel.MIXED:
            # replay the policy's trajectory into the caller's manager (core/precision.py:297-299)
            for code in r["precision_codes"]:
                level = PrecisionLevel.SINGLE if code == 0 else PrecisionLevel.DOUBLE
                if level != pm.current_precision:
                    pm.current_precision = level
                    pm.precision_history.append(level)
        if pm is not None and eng.cfg.precision in (_lib.MG_PREC_ADAPTIVE, _lib.MG_PREC_SINGLE_MANAGED) and r["precision_codes"] and r["precision_codes"][-1] == 0:
            u = u.astype(np.float32)        # the reference returns the fp32 iterate while in SINGLE
        self.level_stats = eng.level_timings() if self.profile else self.level_stats
        self._last = dict(r, wall=wall)
        return u, self.get_convergence_info()

    def get_convergence_info(self):                                          # multigrid.py:377-391
        info = super().get_convergence_info()
        info.update({
            "cycle_type": self.cycle_type,
            "num_levels": len(self.grids),
            "grid_hierarchy": [(g.nx, g.ny) for g in self.grids],
            "level_timings": dict(self.level_stats),
            "pre_smooth_iterations": self.pre_smooth_iterations,
            "post_smooth_iterations": self.post_smooth_iterations,
        })
        last = getattr(self, "_last", None)
        if last:
            info.update({"initial_residual": last["initial_residual"], "device_id": self.device_id,
                         "gpu_solve_time": last["solve_seconds"],
                         "gpu_transfer_time": last["h2d_seconds"] + last["d2h_seconds"],
                         "kernel_time": last["solve_seconds"],
                         # adaptive policy: why the fp32 phase ended ("threshold" / "stagnation" / "fp32_floor"), or
                         # "fp32_skipped" when it was declined a priori (include/mghip.h, mg_stats)
                         "switch_reason": last.get("switch_reason"), "fp32_floor": last.get("fp32_floor", 0.0)})
        return info

    def cleanup(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}


class GPUMultigridSolver(MultigridSolver):
    """gpu/gpu_solver.py:24-501 flavour: string-named smoother, setup() without smoother arguments,
    residual_history that starts with the initial residual, GPU info keys."""

    def __init__(self, device_id=0, max_levels=6, max_iterations=100, tolerance=1e-6, cycle_type="V",
                 pre_smooth_iterations=2, post_smooth_iterations=2, coarse_solver_iterations=10,
                 smoother="jacobi", relaxation_parameter=None, enable_mixed_precision=False,
                 use_tensor_cores=False, memory_pool_size_mb=None):
        super().__init__(max_levels, max_iterations, tolerance, cycle_type, pre_smooth_iterations,
                         post_smooth_iterations, device_id=device_id)
        if smoother not in ("jacobi", "gauss_seidel", "sor"):
            raise ValueError(f"Unknown smoother: {smoother}")
        if _lib.device_count() <= device_id:
            raise RuntimeError(f"mghip: HIP device {device_id} is not available")
        self.smoother_name = smoother
        self.relaxation_parameter = relaxation_parameter
        self.enable_mixed_precision = enable_mixed_precision
        self.coarse_solver_iterations = coarse_solver_iterations      # accepted; the coarsest grid is solved to coarse_tolerance
        self.use_tensor_cores = False                                 # stencils are bandwidth-bound: no matrix-core path

    def setup(self, fine_grid, operator, restriction, prolongation, smoother=None, coarse_solver=None):
        from .smoothers import JacobiSmoother
        if smoother is None:
            if self.smoother_name == "jacobi":
                smoother = JacobiSmoother(relaxation_parameter=self.relaxation_parameter or 0.8)
            else:
                omega = self.relaxation_parameter or (1.0 if self.smoother_name == "gauss_seidel" else 1.15)
                smoother = GaussSeidelSmoother(relaxation_parameter=omega, red_black=True)
        super().setup(fine_grid, operator, restriction, prolongation, smoother, coarse_solver)

    def solve(self, grid, operator, rhs, initial_guess=None, precision_manager=None):
        if precision_manager is None and self.enable_mixed_precision:
            precision_manager = PrecisionManager(default_precision="mixed")
        u, info = super().solve(grid, operator, rhs, initial_guess, precision_manager)
        info["residual_history"] = [info["initial_residual"]] + info["residual_history"]   # gpu_solver.py:246-251,269
        info["smoother"] = self.smoother_name
        info["precision_stats"] = precision_manager.get_statistics() if precision_manager else {}
        self._solves = getattr(self, "_solves", 0) + 1
        self._total_gpu_time = getattr(self, "_total_gpu_time", 0.0) + info.get("gpu_solve_time", 0.0)
        self._total_transfer_time = getattr(self, "_total_transfer_time", 0.0) + info.get("gpu_transfer_time", 0.0)
        return u, info

    def get_performance_statistics(self):                                    # gpu/gpu_solver.py:448-481
        total = getattr(self, "_total_gpu_time", 0.0) + getattr(self, "_total_transfer_time", 0.0)
        return {"device_id": self.device_id, "smoother": self.smoother_name, "solves": getattr(self, "_solves", 0),
                "num_levels": len(self.grids), "grid_hierarchy": [(g.nx, g.ny) for g in self.grids],
                "performance_stats": {"total_gpu_time": getattr(self, "_total_gpu_time", 0.0),
                                      "transfer_time": getattr(self, "_total_transfer_time", 0.0),
                                      "kernel_time": getattr(self, "_total_gpu_time", 0.0)},
                "gpu_utilization": (getattr(self, "_total_gpu_time", 0.0) / total) if total > 0 else 0.0}


class GPUCommunicationAvoidingMultigrid(GPUMultigridSolver):
    """gpu/gpu_solver.py:504-798 flavour: GPUMultigridSolver + a full-multigrid start (`use_fmg`, `fmg_cycles`) and the
    bookkeeping keys its callers read (applications/poisson_solver.py:90-101 builds it with use_fmg=True).

    What the reference means by "communication avoiding" -- block-structured smoothing through shared memory, fused
    residual + restriction, fused prolongation + correction, asynchronous streams, a device memory pool -- is how EVERY
    cycle of this engine runs (fused legs, LDS / register tiles, one stream, one arena allocated at setup), so block_size,
    enable_memory_pool and async_operations are accepted and reported but select nothing.  FMG: the reference restricts
    the rhs, solves the coarsest level and prolongs upward with `fmg_cycles` x pre-smoothing per level
    (gpu_solver.py:603-652); here every level gets `fmg_cycles` full cycles of its sub-hierarchy (mg_fmg, include/mghip.h),
    the textbook form -- only used when no initial guess is given, like the reference (:583)."""

    def __init__(self, device_id=0, max_levels=6, cycle_type="V", block_size=32, enable_memory_pool=True, use_fmg=False,
                 fmg_cycles=1, async_operations=True, **kwargs):
        super().__init__(device_id=device_id, max_levels=max_levels, cycle_type=cycle_type, **kwargs)
        self.block_size = block_size
        self.enable_memory_pool = enable_memory_pool
        self.use_fmg = bool(use_fmg)
        self.async_operations = async_operations
        self.fmg_cycles = int(fmg_cycles) if use_fmg else 0          # MultigridSolver._engine passes it to mg_config.fmg_cycles
        self.ca_stats = {"block_operations": 0, "async_operations": 0, "fmg_initializations": 0, "memory_pool_hits": 0}
        self.name = f"GPU-CA-MG-{cycle_type}cycle"

    def solve(self, grid, operator, rhs, initial_guess=None, precision_manager=None):
        u, info = super().solve(grid, operator, rhs, initial_guess, precision_manager)
        n = info["iterations"]
        legs = 2 * max(len(self.grids) - 1, 0)                     # fused down + up leg per level and cycle
        self.ca_stats["block_operations"] += n * legs
        self.ca_stats["async_operations"] += n if self.async_operations else 0
        if self.use_fmg and initial_guess is None:
            self.ca_stats["fmg_initializations"] += 1
        info.update({"ca_optimizations": True, "block_size": self.block_size, "fmg_used": self.use_fmg,
                     "async_operations": self.async_operations, "ca_stats": self.ca_stats.copy()})
        return u, info

    def get_performance_statistics(self):
        stats = super().get_performance_statistics()
        stats["ca_optimizations"] = {"block_size": self.block_size, "ca_stats": self.ca_stats.copy(),
                                     "memory_pool_enabled": self.enable_memory_pool, "fmg_enabled": self.use_fmg,
                                     "async_enabled": self.async_operations}
        return stats
