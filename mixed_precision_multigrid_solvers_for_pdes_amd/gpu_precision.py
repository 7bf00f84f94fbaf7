"""Device-side precision policy.  Mirrors multigrid.gpu.gpu_precision (gpu/gpu_precision.py:20-419): GPUPrecisionLevel and
GPUPrecisionManager with the same constructor, thresholds (1e-2 / 1e-8 / 4096 MB), method names and statistics keys.

Policy only (SURVEY section 2, row gpu_precision.py): the multigrid path computes in fp32 or fp64.  The reference's
`MIXED_TC` default means "fp32 on level 0, fp16 on the coarser levels when tensor cores exist, else fp32"
(gpu_precision.py:116-142) and its "tensor-core optimisation" is an element-wise fp16 add (:176-227) -- a 5-point stencil has
nothing for a matrix core to do, so `tensor_core_available` is False on this engine and every rule below takes the branch the
reference takes without tensor cores (fp32 where it would use fp16).  `to_engine_policy()` maps a manager onto the engine's
policies (mg_config.precision): that is how `GPUMultigridSolver(enable_mixed_precision=True)` and the facade consume it."""
import time
from enum import Enum

import numpy as np

from . import _lib


class GPUPrecisionLevel(Enum):                                           # gpu/gpu_precision.py:20-25
    HALF = "half"
    SINGLE = "single"
    DOUBLE = "double"
    MIXED_TC = "mixed_tc"


class GPUPrecisionManager:
    def __init__(self, device_id=0, enable_tensor_cores=True, adaptive=True, default_precision="mixed_tc"):
        _lib.load()                                                          # the reference raises ImportError without its backend (:50-51)
        self.device_id = device_id
        self.enable_tensor_cores = enable_tensor_cores
        self.adaptive = adaptive
        values = [lvl.value for lvl in GPUPrecisionLevel]
        self.current_precision = GPUPrecisionLevel(default_precision) if default_precision in values else GPUPrecisionLevel.MIXED_TC
        self.thresholds = {"downgrade_residual": 1e-2, "upgrade_residual": 1e-8, "memory_pressure_mb": 4096}   # :65-69
        self.precision_history = []
        self.performance_stats = {"fp16_operations": 0, "fp32_operations": 0, "fp64_operations": 0,
                                  "tensor_core_operations": 0, "precision_switches": 0}
        self.tensor_core_available = self._check_tensor_core_support()

    def _check_tensor_core_support(self):
        """Stencils are bandwidth-bound (about 10 flop per 12-24 B): there is no matrix-core path on this engine."""
        return False

    def get_optimal_dtype(self, operation_type, grid_level=0):              # gpu/gpu_precision.py:116-142, no-tensor-core branch
        if self.current_precision == GPUPrecisionLevel.DOUBLE:
            return np.float64
        # HALF has no arithmetic on the multigrid path: the narrowest working precision is fp32
        return np.float32

    def convert_to_optimal_precision(self, array, operation_type, grid_level=0):   # :144-174 (NumPy arrays and torch tensors)
        dt = np.dtype(self.get_optimal_dtype(operation_type, grid_level))
        is_np = isinstance(array, np.ndarray)
        cur = array.dtype if is_np else np.dtype(str(array.dtype).replace("torch.", ""))
        if cur == dt:
            return array
        self.performance_stats["fp32_operations" if dt == np.float32 else "fp64_operations"] += 1
        if is_np:
            return array.astype(dt)
        import torch
        return array.to(torch.float32 if dt == np.float32 else torch.float64)

    def apply_tensor_core_optimization(self, array1, array2, operation="multiply"):   # :176-227, the fall-back branch
        if operation == "multiply":
            return array1 * array2
        if operation == "add":
            return array1 + array2
        return array1

    def update_precision_adaptive(self, residual_norm, grid_shapes=None, memory_usage_mb=None):   # :229-289
        if not self.adaptive:
            return False
        old = self.current_precision
        changed = False
        if memory_usage_mb and memory_usage_mb > self.thresholds["memory_pressure_mb"]:
            if self.current_precision in (GPUPrecisionLevel.DOUBLE, GPUPrecisionLevel.SINGLE):
                self.current_precision = GPUPrecisionLevel.SINGLE
                changed = self.current_precision != old
        if residual_norm > self.thresholds["downgrade_residual"]:
            if self.current_precision == GPUPrecisionLevel.DOUBLE:
                self.current_precision = GPUPrecisionLevel.SINGLE
                changed = True
        elif residual_norm < self.thresholds["upgrade_residual"]:
            if self.current_precision in (GPUPrecisionLevel.HALF, GPUPrecisionLevel.MIXED_TC):
                self.current_precision = GPUPrecisionLevel.SINGLE
                changed = True
        if changed:
            self.performance_stats["precision_switches"] += 1
            self.precision_history.append({"timestamp": time.time(), "old_precision": old.value,
                                           "new_precision": self.current_precision.value, "residual_norm": residual_norm,
                                           "memory_usage_mb": memory_usage_mb, "reason": "adaptive_switch"})
        return changed

    def get_precision_recommendations(self, grid_hierarchy_info):           # :291-323, no-tensor-core / no-fp16 branch
        n = grid_hierarchy_info.get("num_levels", 1)
        return {level: (self.current_precision.value if level == 0 else GPUPrecisionLevel.SINGLE.value) for level in range(n)}

    def create_mixed_precision_arrays(self, base_array, operation_types, grid_levels):   # :325-353
        return {f"{op}_level_{lvl}": self.convert_to_optimal_precision(base_array, op, lvl)
                for op, lvl in zip(operation_types, grid_levels)}

    def estimate_speedup(self, operation_type, array_size):
        """Traffic ratio of the working precision against fp64 (the legs are HBM-bound): 2.0 in fp32, 1.0 in fp64.  The
        reference returns hard-coded guesses here (:355-384)."""
        return 1.0 if self.current_precision == GPUPrecisionLevel.DOUBLE else 2.0

    def get_precision_statistics(self):                                     # :386-406
        s = self.performance_stats
        total = s["fp16_operations"] + s["fp32_operations"] + s["fp64_operations"]
        return {"current_precision": self.current_precision.value, "tensor_core_available": self.tensor_core_available,
                "performance_stats": s.copy(),
                "precision_distribution": {"fp16_percent": s["fp16_operations"] / max(total, 1) * 100,
                                           "fp32_percent": s["fp32_operations"] / max(total, 1) * 100,
                                           "fp64_percent": s["fp64_operations"] / max(total, 1) * 100},
                "precision_switches": s["precision_switches"], "tensor_core_utilization": s["tensor_core_operations"],
                "adaptive_enabled": self.adaptive}

    get_statistics = get_precision_statistics                               # what GPUMultigridSolver.solve reports as precision_stats

    def reset_statistics(self):                                             # :408-418
        for k in self.performance_stats:
            self.performance_stats[k] = 0
        self.precision_history.clear()

    # ---- bridge to the engine ---------------------------------------------------------------------------------------
    def to_engine_policy(self, switch_threshold=1e-6):
        """core.PrecisionManager equivalent the solver drivers translate into mg_config.precision: DOUBLE -> fp64 levels;
        SINGLE / HALF -> fp32 levels with an fp64 coarsest solve; MIXED_TC -> adaptive fp32 -> fp64 (one-way, promotion at
        10 x switch_threshold or on stagnation) when `adaptive`, else per-level mixed."""
        from .precision import PrecisionManager
        if self.current_precision == GPUPrecisionLevel.DOUBLE:
            return PrecisionManager("double", adaptive=False, convergence_threshold=switch_threshold)
        if self.current_precision in (GPUPrecisionLevel.SINGLE, GPUPrecisionLevel.HALF):
            return PrecisionManager("single", adaptive=False, convergence_threshold=switch_threshold)
        if self.adaptive:
            pm = PrecisionManager("double", adaptive=True, convergence_threshold=switch_threshold)
            pm.reference_rule = False
            return pm
        return PrecisionManager("mixed", adaptive=True, convergence_threshold=switch_threshold)
