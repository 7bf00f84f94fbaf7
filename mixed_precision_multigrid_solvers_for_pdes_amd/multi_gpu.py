"""The decomposed engine behind the reference's multi-GPU solver API.

Mirrors multigrid.gpu.multi_gpu.DistributedMultigridSolver (gpu/multi_gpu.py:301-750: `setup(global_grid, operator,
restriction, prolongation)`, `solve(global_grid, operator, global_rhs, initial_guess=None, precision_manager=None) ->
(u, info)`, `get_performance_statistics()`, `cleanup()`) and multigrid.gpu.multi_gpu_solver.MultiGPUSolver
(gpu/multi_gpu_solver.py:188-340: `domain_decomposition_solve(grid, rhs, initial_guess) -> dict`), plus MultiGPUManager /
DecompositionType as the names their callers import.

What is kept from the reference is the interface and the partitioning vocabulary ("stripe" = rows cut into N stripes,
"checkerboard" = a square process grid, 1-cell overlap: gpu/multi_gpu.py:386-476).  Its algorithm is not: it solves every
sub-domain INDEPENDENTLY and stitches the pieces (gpu/multi_gpu.py:540-607), respectively replaces the coarse-grid
correction by `u += 0.8 r` (gpu/multi_gpu_solver.py:574-593) -- neither is a multigrid solve of the global problem (SURVEY
F6).  Here the SAME V/W/F-cycle as the single-GPU engine runs on a px x py block decomposition (distributed.py: fused legs
on ghost zones, RCCL halo exchange, agglomerated coarse levels, one all-reduce for the norm), so `solve` returns the
single-GPU solver's iterate bit for bit, with the same residual history and info keys, plus `n_gpus`, `process_grid`,
`exchanges_per_cycle` and the reference's aggregate keys.

Where the ranks live:
  * under torch.distributed (one process per GPU, `python -m torch.distributed.run ...`, backend "nccl" = RCCL): every rank
    constructs the solver and calls setup / solve with the same global arrays (SPMD); every rank gets the global solution;
  * otherwise: `len(device_ids)` sub-domains as virtual ranks inside this process, all on `device_ids[0]` (a one-GPU box:
    tests, rehearsals; one HIP context drives one GPU at full rate, a second device needs a second process).
There is no CPU fallback: without a usable device the constructor raises like the reference's does without CuPy.
"""
import logging
import os
import time
from enum import Enum

import numpy as np

from . import _lib
from . import distributed as D
from .precision import PrecisionLevel
from .smoothers import BaseSolver
from .solver import GPUMultigridSolver, _precision_config

logger = logging.getLogger(__name__)


class DecompositionType(Enum):                                           # gpu/multi_gpu_solver.py:22-28
    STRIP_X = "strip_x"
    STRIP_Y = "strip_y"
    BLOCK_2D = "block_2d"
    ADAPTIVE = "adaptive"


def _dist_if_initialised():
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def process_grid_for(strategy, n):
    """(px, py) of a decomposition strategy: the reference's "stripe" cuts rows (gpu/multi_gpu.py:390-430), its
    "checkerboard" needs a square number of devices (:432-436); "block" / "adaptive" = as square as possible."""
    s = strategy.value if isinstance(strategy, DecompositionType) else str(strategy)
    if s in ("stripe", "strip_x"):
        return n, 1
    if s == "strip_y":
        return 1, n
    if s == "checkerboard":
        k = int(round(np.sqrt(n)))
        if k * k != n:
            raise ValueError("Checkerboard decomposition requires square number of devices")
        return k, k
    if s in ("block", "block_2d", "adaptive", "auto"):
        return D.process_grid(n)
    raise ValueError(f"Unknown decomposition strategy: {strategy}")


class MultiGPUManager:
    """gpu/multi_gpu.py:39-224: which devices take part.  Memory, streams and load balancing are the engine's (one arena per
    handle, equal blocks per rank), so this is device discovery and bookkeeping only."""

    def __init__(self, device_ids=None):
        _lib.load()
        n = _lib.device_count()
        if n <= 0:
            raise RuntimeError("mghip: no usable HIP device (the multi-GPU path has no CPU fallback)")
        self.available_devices = self._discover_gpus(n)
        self.device_ids = list(range(n)) if device_ids is None else list(device_ids)
        for d in self.device_ids:
            if d >= n or d < 0:
                raise ValueError(f"Device ID {d} not available")
        self.num_devices = len(self.device_ids)
        self.active_tasks = {d: [] for d in set(self.device_ids)}

    @staticmethod
    def _discover_gpus(n):
        import torch
        out = []
        for d in range(n):
            p = torch.cuda.get_device_properties(d) if d < torch.cuda.device_count() else None
            out.append({"device_id": d, "name": getattr(p, "name", "gfx950"), "total_memory": getattr(p, "total_memory", 0),
                        "multiprocessor_count": getattr(p, "multi_processor_count", 0)})
        return out

    def get_optimal_device(self, memory_requirement_mb=0):
        return min(self.active_tasks, key=lambda d: len(self.active_tasks[d]))

    def allocate_device_for_task(self, task_name, memory_requirement_mb=0, preferred_device=None):
        d = preferred_device if preferred_device in self.active_tasks else self.get_optimal_device(memory_requirement_mb)
        self.active_tasks[d].append(task_name)
        return d

    def release_device(self, device_id, task_name):
        if task_name in self.active_tasks.get(device_id, []):
            self.active_tasks[device_id].remove(task_name)

    def get_device_status(self):
        return {d: dict(self.available_devices[d], active_tasks=list(t)) for d, t in self.active_tasks.items()}

    def cleanup(self):
        for t in self.active_tasks.values():
            t.clear()


class DistributedMultigridSolver(BaseSolver):
    """See the module docstring.  `solver_kwargs` are GPUMultigridSolver's (gpu/gpu_solver.py:32-46): max_levels,
    max_iterations, tolerance, cycle_type, pre_/post_smooth_iterations, smoother in {'jacobi', 'gauss_seidel', 'sor'},
    relaxation_parameter, enable_mixed_precision, ... plus coarse_tolerance / coarse_max_iterations.

    agglomerate_at: levels of at most this many points per direction run replicated on every GPU after one all-gather.
    ops_factory(dtype, managed_single=False, mixed=False): kernel provider (default: libmghip's device kernels, HipOps);
    the CPU tests inject their NumPy stand-in here, everything else is the code a GPU run executes."""

    def __init__(self, device_ids=None, decomposition_strategy="stripe", communication_method="p2p", agglomerate_at=1025,
                 ops_factory=None, **solver_kwargs):
        kw = dict(solver_kwargs)
        super().__init__(kw.get("max_iterations", 100), kw.get("tolerance", 1e-6), False, "DistributedMultigrid")
        self.dist = _dist_if_initialised()
        self._ops_factory = ops_factory
        if ops_factory is None:
            self.multi_gpu_manager = MultiGPUManager(None if self.dist is not None else device_ids)
        else:
            self.multi_gpu_manager = None
        if self.dist is not None:
            self.world, self.rank = self.dist.get_world_size(), self.dist.get_rank()
            local = int(os.environ.get("LOCAL_RANK", self.rank))
            self.device_ids = list(device_ids) if device_ids is not None else list(range(self.world))
            self.device = self.device_ids[self.rank] if device_ids is not None else local
            self.ranks = [self.rank]
        else:
            self.device_ids = list(device_ids) if device_ids is not None else [0]
            self.world, self.rank = len(self.device_ids), 0
            self.device = self.device_ids[0]
            self.ranks = list(range(self.world))
        self.num_devices = self.world
        self.decomposition_strategy = decomposition_strategy
        self.communication_method = communication_method
        self.px, self.py = process_grid_for(decomposition_strategy, self.world)
        self.agglomerate_at = agglomerate_at
        self.solver_kwargs = kw
        self.max_levels = kw.get("max_levels", 6)
        self.cycle_type = kw.get("cycle_type", "V")
        self.pre = kw.get("pre_smooth_iterations", 2)
        self.post = kw.get("post_smooth_iterations", 2)
        self.coarse_tolerance = kw.get("coarse_tolerance", 1e-12)
        self.coarse_max_iterations = kw.get("coarse_max_iterations", 1000)
        self.enable_mixed_precision = kw.get("enable_mixed_precision", False)
        name = kw.get("smoother", "jacobi")
        if name not in ("jacobi", "gauss_seidel", "sor"):
            raise ValueError(f"Unknown smoother: {name}")
        self.smoother_name = name
        omega = kw.get("relaxation_parameter")
        self.smoother = "jacobi" if name == "jacobi" else "rbgs"
        self.omega = omega or (0.8 if name == "jacobi" else (1.0 if name == "gauss_seidel" else 1.15))
        self.global_grid = None
        self.coeff = -1.0
        self.subdomain_info, self.communication_graph = {}, {}
        self._loops = {}
        self._single = None               # the single-GPU solver, when no level of the grid can be decomposed
        self._last = {}

    # -- setup (gpu/multi_gpu.py:348-384) ------------------------------------------------------------------------------
    def setup(self, global_grid, operator, restriction, prolongation):
        if getattr(restriction, "method", None) != "full_weighting" or getattr(prolongation, "method", None) != "bilinear":
            raise NotImplementedError("the accelerated path implements full_weighting restriction and bilinear prolongation")
        if hasattr(operator, "field") or getattr(operator, "shift", 0.0):
            raise NotImplementedError("DistributedMultigridSolver runs the constant-coefficient operator; the variable-coefficient "
                                      "decomposed cycle is DistributedMultigrid.set_coefficient (distributed.py)")
        self.cleanup()
        self.global_grid = global_grid
        self.coeff = float(getattr(operator, "coefficient", -1.0))
        self.shapes = D.hierarchy_shapes(global_grid.nx, global_grid.ny, self.max_levels)
        G = D.GHOST_FUSED[self.smoother] if (self.pre <= 2 and self.post <= 2) else 1
        self.Ld = D.distributed_levels(self.shapes, self.px, self.py, self.agglomerate_at, G) if self.world > 1 else 0
        self._decompose_domain(global_grid)
        self._setup_communication()
        if self.Ld == 0:
            # no level can be cut px x py (too small, or n - 1 not divisible): every rank runs the single-GPU engine on the
            # whole grid -- same answer, no exchange
            accepted = ("max_levels", "max_iterations", "tolerance", "cycle_type", "pre_smooth_iterations", "post_smooth_iterations",
                        "coarse_solver_iterations", "smoother", "relaxation_parameter", "enable_mixed_precision", "use_tensor_cores",
                        "memory_pool_size_mb")
            kw = {k: v for k, v in self.solver_kwargs.items() if k in accepted}
            self._single = GPUMultigridSolver(device_id=self.device, **kw)
            self._single.coarse_tolerance, self._single.coarse_max_iterations = self.coarse_tolerance, self.coarse_max_iterations
            self._single.setup(global_grid, operator, restriction, prolongation)
            return
        self._loop(global_grid, None)      # build the default hierarchy now so that setup fails early, like the reference

    def _decompose_domain(self, grid):                                     # gpu/multi_gpu.py:386-476 (bookkeeping of OUR blocks)
        self.subdomain_info = {}
        for r in range(self.world):
            rx, ry = divmod(r, self.py)
            b = D.Block(grid.nx, grid.ny, self.px, self.py, rx, ry, 1) if self.Ld > 0 else None
            if b is None:
                self.subdomain_info[r] = {"global_slice": (slice(0, grid.nx), slice(0, grid.ny)), "row_idx": rx, "col_idx": ry, "replicated": True}
                continue
            self.subdomain_info[r] = {"global_slice": (slice(b.gx0 + b.i_lo, b.gx0 + b.i_hi), slice(b.gy0 + b.j_lo, b.gy0 + b.j_hi)),
                                      "row_idx": rx, "col_idx": ry, "has_top_boundary": rx == 0, "has_bottom_boundary": rx == self.px - 1,
                                      "replicated": False}

    def _setup_communication(self):                                        # gpu/multi_gpu.py:478-517 (+ the diagonal neighbours)
        self.communication_graph = {}
        for r in range(self.world):
            rx, ry = divmod(r, self.py)
            self.communication_graph[r] = [qx * self.py + qy for qx in (rx - 1, rx, rx + 1) for qy in (ry - 1, ry, ry + 1)
                                           if (qx, qy) != (rx, ry) and 0 <= qx < self.px and 0 <= qy < self.py]

    # -- one DecomposedSolve per precision policy ----------------------------------------------------------------------
    def _ops(self, dtype, **kw):
        if self._ops_factory is not None:
            return self._ops_factory(dtype, **kw)
        import torch
        torch.cuda.set_device(self.device)
        return D.HipOps(dtype, torch.device("cuda", self.device), **kw)

    def _dm(self, ops):
        g = self.global_grid
        return D.DistributedMultigrid(g.nx, g.ny, self.px, self.py, self.ranks, ops, self.dist, domain=g.domain, coeff=self.coeff,
                                      max_levels=self.max_levels, cycle=self.cycle_type, pre=self.pre, post=self.post,
                                      smoother=self.smoother, omega=self.omega, coarse_tol=self.coarse_tolerance,
                                      coarse_maxit=self.coarse_max_iterations, agglomerate_at=self.agglomerate_at)

    def _loop(self, grid, pm):
        prec, thr, _mem, ref_rule = key = _precision_config(grid, pm)
        if key in self._loops:
            return self._loops[key]
        if prec == _lib.MG_PREC_DOUBLE:
            loop = D.DecomposedSolve({"f64": self._dm(self._ops(np.float64))})
        elif prec == _lib.MG_PREC_SINGLE:
            loop = D.DecomposedSolve({"f32": self._dm(self._ops(np.float32))})
        elif prec == _lib.MG_PREC_SINGLE_MANAGED:
            loop = D.DecomposedSolve({"f32": self._dm(self._ops(np.float32, managed_single=True))})
        elif prec == _lib.MG_PREC_MIXED_LEVELS:
            loop = D.DecomposedSolve({"f64": self._dm(self._ops(np.float64, mixed=True))})
        elif prec == _lib.MG_PREC_ADAPTIVE and not ref_rule:
            loop = D.DecomposedSolve({"f32": self._dm(self._ops(np.float32, managed_single=True)),
                                      "f64": self._dm(self._ops(np.float64))}, "adaptive", thr)
        elif prec == _lib.MG_PREC_ADAPTIVE:
            raise NotImplementedError("the reference's two-way adaptive rule never recovers from fp32 (SURVEY F11); the decomposed "
                                      "solver runs the one-way rule: set precision_manager.reference_rule = False")
        else:
            raise NotImplementedError("defect correction is a single-GPU policy (mg_config.precision = MG_PREC_DEFECT)")
        self._loops[key] = loop
        return loop

    # -- solve (gpu/multi_gpu.py:540-607) -----------------------------------------------------------------------------
    def solve(self, global_grid, operator, global_rhs, initial_guess=None, precision_manager=None):
        if self.global_grid is None or global_grid.shape != self.global_grid.shape:
            raise ValueError("Multigrid not properly setup or grid mismatch")
        pm = precision_manager
        if pm is None and self.enable_mixed_precision:
            from .precision import PrecisionManager
            pm = PrecisionManager(default_precision="mixed")
        t_start = time.time()
        if self._single is not None:
            u, info = self._single.solve(global_grid, operator, global_rhs, initial_guess, pm)
            info["residual_history"] = info["residual_history"][1:]          # the CPU flavour's history (no initial residual)
            return u, self._finish_info(info, time.time() - t_start, decomposed=False)
        self.reset()
        rhs = _lib.as_c(global_rhs)
        if rhs.shape != global_grid.shape:
            raise ValueError(f"Field shape {rhs.shape} doesn't match grid shape {global_grid.shape}")
        u0 = None if initial_guess is None else _lib.as_c(initial_guess)
        loop = self._loop(global_grid, pm)
        for sv in loop.solvers.values():
            sv.exchanges = 0
            sv.native_cycles = 0
        cut = lambda a: (lambda b: a[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
        t0 = time.time()
        r0 = loop.set_problem(cut(rhs), None if u0 is None else cut(u0))       # block scatter (gpu/multi_gpu.py:609-627)
        t_scatter = time.time() - t0
        t0 = time.time()
        hist, phases, conv = loop.run(self.tolerance, self.max_iterations)
        t_solve = time.time() - t0
        t0 = time.time()
        out_dtype = np.float32 if (np.dtype(global_grid.dtype) == np.float32 and rhs.dtype == np.float32) else np.float64
        if loop.policy_kind == "adaptive" and phases and phases[-1] == "f32":
            out_dtype = np.float32                                             # the reference returns the fp32 iterate while in SINGLE
        u = self._gather_solution(loop.current, out_dtype)                     # gpu/multi_gpu.py:629-667
        t_gather = time.time() - t0
        n = len(hist)
        names = {"f32": "float32", "f64": "float64"}
        mixed = any(getattr(sv, "mixed", False) for sv in loop.solvers.values())
        for k in range(n):
            level = ("mixed" if mixed else names[phases[k]]) if pm is not None else "double"
            self.history.record_iteration(hist[k], t_solve / max(n, 1), level, 0)
        self.converged, self.iterations_performed = conv, n
        self.final_residual = hist[-1] if n else r0
        if not conv:
            logger.warning(f"{self.name} reached max iterations ({self.max_iterations}): residual = {self.final_residual:.2e}")
        if pm is not None and pm.adaptive and pm.current_precision != PrecisionLevel.MIXED:
            for ph in phases:                                                  # the policy's trajectory into the caller's manager
                lvl = PrecisionLevel.SINGLE if ph == "f32" else PrecisionLevel.DOUBLE
                if lvl != pm.current_precision:
                    pm.current_precision = lvl
                    pm.precision_history.append(lvl)
        sv0 = loop.current
        info = self.get_convergence_info()
        info.update({"cycle_type": self.cycle_type, "num_levels": sv0.L, "grid_hierarchy": list(sv0.shapes),
                     "level_timings": {}, "pre_smooth_iterations": self.pre, "post_smooth_iterations": self.post,
                     "initial_residual": r0, "device_id": self.device, "smoother": self.smoother_name,
                     "gpu_solve_time": t_solve, "gpu_transfer_time": t_scatter + t_gather, "kernel_time": t_solve,
                     "precision_stats": pm.get_statistics() if pm is not None else {},
                     "precision_switches": loop.switches,
                     "exchanges_per_cycle": sum(x.exchanges for x in loop.solvers.values()) / max(n, 1),
                     "distributed_levels": sv0.Ld, "ghost_width": sv0.G, "mode": sv0.mode,
                     "native_plan_cycles": sum(x.native_cycles for x in loop.solvers.values())})
        self._last = info
        return u, self._finish_info(info, time.time() - t_start, decomposed=True)

    def _finish_info(self, info, total, decomposed):
        px, py = (self.px, self.py) if decomposed else (1, 1)
        info.update({"n_gpus": self.world, "num_devices": self.world, "process_grid": (px, py), "decomposed": decomposed,
                     "decomposition_strategy": self.decomposition_strategy, "communication_method": self.communication_method,
                     "total_iterations": info["iterations"], "average_iterations": float(info["iterations"]),
                     "distributed_solve_time": total,
                     "device_stats": {r: {"iterations": info["iterations"], "final_residual": info["final_residual"],
                                          "converged": info["converged"], "solve_time": info.get("gpu_solve_time", 0.0)}
                                      for r in range(self.world)}})
        info.setdefault("exchanges_per_cycle", 0.0)
        return info

    def _gather_solution(self, sv, dtype):
        """every rank's exclusive window -> the global array (on every rank)"""
        g = self.global_grid
        out = np.empty((g.nx, g.ny), dtype=dtype)
        pieces = []
        for r in sv.ranks:
            b, u = sv.local_solution(r)
            pieces.append((b.gx0 + b.i_lo, b.gx0 + b.i_hi, b.gy0 + b.j_lo, b.gy0 + b.j_hi,
                           np.ascontiguousarray(u[b.i_lo:b.i_hi, b.j_lo:b.j_hi])))
        if self.dist is not None:
            every = [None] * self.world
            self.dist.all_gather_object(every, pieces)
            pieces = [p for part in every for p in part]
        for i0, i1, j0, j1, a in pieces:
            out[i0:i1, j0:j1] = a
        return out

    def get_performance_statistics(self):                                   # gpu/multi_gpu.py:704-730
        return {"multi_gpu_stats": {"num_devices": self.num_devices, "decomposition_strategy": self.decomposition_strategy,
                                    "communication_method": self.communication_method, "process_grid": (self.px, self.py)},
                "device_statistics": dict(self._last.get("device_stats", {})),
                "load_balancer_stats": {"device_loads": {r: 1.0 / self.world for r in range(self.world)},
                                        "active_tasks": {r: 0 for r in range(self.world)}}}

    def cleanup(self):
        for loop in self._loops.values():
            loop.close()
        self._loops = {}
        if self._single is not None:
            self._single.cleanup()
            self._single = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


class MultiGPUSolver:
    """gpu/multi_gpu_solver.py:188-340: `domain_decomposition_solve(grid, rhs, initial_guess) -> dict` on the same decomposed
    engine (the reference's own loop there smooths sub-domains and adds 0.8 r as its "coarse correction")."""

    def __init__(self, num_gpus, decomposition_type=DecompositionType.ADAPTIVE, max_levels=4, max_iterations=50, tolerance=1e-10,
                 load_balance_threshold=0.1, **solver_kwargs):
        self.num_gpus = int(num_gpus)
        self.decomposition_type = decomposition_type
        self.max_levels, self.max_iterations, self.tolerance = max_levels, max_iterations, tolerance
        self.load_balance_threshold = load_balance_threshold
        self._kwargs = solver_kwargs
        self._solver = None

    def domain_decomposition_solve(self, grid, rhs, initial_guess=None, precision_managers=None):
        from .operators import LaplacianOperator, ProlongationOperator, RestrictionOperator
        t0 = time.time()
        if self._solver is None or self._solver.global_grid is None or self._solver.global_grid.shape != grid.shape:
            if self._solver is not None:
                self._solver.cleanup()
            self._solver = DistributedMultigridSolver(device_ids=[0] * self.num_gpus if _dist_if_initialised() is None else None,
                                                      decomposition_strategy=self.decomposition_type, max_levels=self.max_levels,
                                                      max_iterations=self.max_iterations, tolerance=self.tolerance, **self._kwargs)
            self._solver.setup(grid, LaplacianOperator(coefficient=-1.0), RestrictionOperator("full_weighting"),
                               ProlongationOperator("bilinear"))
        pm = precision_managers[0] if precision_managers else None
        u, info = self._solver.solve(grid, self._solver_operator(), rhs, initial_guess, pm)
        return {"solution": u, "converged": info["converged"], "iterations": info["iterations"],
                "final_residual": info["final_residual"], "residual_history": info["residual_history"],
                "solve_time": time.time() - t0, "num_gpus_used": self.num_gpus,
                "domain_decomposition": getattr(self.decomposition_type, "value", self.decomposition_type),
                "performance_stats": info["device_stats"], "info": info}

    @staticmethod
    def _solver_operator():
        from .operators import LaplacianOperator
        return LaplacianOperator(coefficient=-1.0)

    def cleanup(self):
        if self._solver is not None:
            self._solver.cleanup()
            self._solver = None
