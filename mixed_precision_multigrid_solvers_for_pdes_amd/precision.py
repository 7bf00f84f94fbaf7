"""Precision policy (host side).  Mirrors multigrid.core.precision (core/precision.py:11-417):
PrecisionLevel and PrecisionManager with the same thresholds and rules.  The policy is pure
bookkeeping; the casts it decides on happen in device kernels (mg_config.precision)."""
from enum import Enum

import numpy as np


class PrecisionLevel(Enum):
    SINGLE = "float32"
    DOUBLE = "float64"
    MIXED = "mixed"


class PrecisionManager:
    def __init__(self, default_precision=PrecisionLevel.DOUBLE, adaptive=True,
                 convergence_threshold=1e-6, memory_threshold_gb=4.0):
        self.default_precision = self._parse_precision(default_precision)
        self.adaptive = adaptive
        self.convergence_threshold = convergence_threshold
        self.memory_threshold_bytes = memory_threshold_gb * 1024**3
        self.precision_hierarchy = [PrecisionLevel.SINGLE, PrecisionLevel.DOUBLE]
        self.current_precision = self.default_precision
        self.precision_history = [self.current_precision]
        self.precision_stats = {PrecisionLevel.SINGLE: {"operations": 0, "time": 0.0},
                                PrecisionLevel.DOUBLE: {"operations": 0, "time": 0.0}}

    @property
    def memory_threshold_gb(self):
        return self.memory_threshold_bytes / 1024**3

    def _parse_precision(self, precision):                                   # core/precision.py:66-83
        if isinstance(precision, str):
            table = {"single": PrecisionLevel.SINGLE, "double": PrecisionLevel.DOUBLE,
                     "mixed": PrecisionLevel.MIXED, "float32": PrecisionLevel.SINGLE,
                     "float64": PrecisionLevel.DOUBLE}
            if precision.lower() in table:
                return table[precision.lower()]
            raise ValueError(f"Unknown precision level: {precision}")
        if isinstance(precision, PrecisionLevel):
            return precision
        raise TypeError(f"Precision must be PrecisionLevel or str, got {type(precision)}")

    def get_dtype(self, precision=None):                                     # core/precision.py:85-104
        precision = self.current_precision if precision is None else precision
        return np.float32 if precision == PrecisionLevel.SINGLE else np.float64

    def convert_array(self, array, target_precision=None):                   # core/precision.py:106-134
        dt = self.get_dtype(target_precision)
        return array if array.dtype == dt else array.astype(dt)

    def estimate_memory_usage(self, grid_shapes):                            # core/precision.py:136-153
        return sum(a * b for a, b in grid_shapes) * np.dtype(self.get_dtype()).itemsize * 4

    def should_downgrade_precision(self, grid_shapes, residual_norm):        # core/precision.py:155-187
        if not self.adaptive:
            return False
        if self.estimate_memory_usage(grid_shapes) > self.memory_threshold_bytes:
            return True
        return (self.current_precision == PrecisionLevel.DOUBLE and
                residual_norm > self.convergence_threshold * 100)

    def should_promote_precision(self, convergence_history, current_precision):   # core/precision.py:189-246
        if not self.adaptive or current_precision == PrecisionLevel.DOUBLE:
            return False
        if len(convergence_history) < 5:
            return False
        r = convergence_history[-5:]
        ratios = [r[i] / r[i - 1] for i in range(1, len(r)) if r[i - 1] > 0]
        if ratios:
            if np.mean(ratios) > 0.9:
                return True
            rel = [abs(r[i] - r[i - 1]) / r[i - 1] for i in range(1, len(r)) if r[i - 1] > 0]
            if rel and np.mean(rel) < 1e-3:
                return True
        return all(r[i] >= r[i - 1] * 0.99 for i in range(1, len(r)))

    def should_upgrade_precision(self, residual_norm):                       # core/precision.py:248-268
        return bool(self.adaptive and self.current_precision == PrecisionLevel.SINGLE and
                    residual_norm < self.convergence_threshold * 10)

    def update_precision(self, residual_norm, grid_shapes=None):             # core/precision.py:270-302
        if not self.adaptive:
            return False
        old = self.current_precision
        if grid_shapes and self.should_downgrade_precision(grid_shapes, residual_norm):
            if self.current_precision == PrecisionLevel.DOUBLE:
                self.current_precision = PrecisionLevel.SINGLE
        elif self.should_upgrade_precision(residual_norm):
            if self.current_precision == PrecisionLevel.SINGLE:
                self.current_precision = PrecisionLevel.DOUBLE
        if self.current_precision != old:
            self.precision_history.append(self.current_precision)
            return True
        return False

    def optimal_precision_per_level(self, grid_level, problem_size):         # core/precision.py:304-335
        if not self.adaptive:
            return self.current_precision
        if grid_level == 0:
            return PrecisionLevel.DOUBLE
        if grid_level <= 2:
            return PrecisionLevel.SINGLE if problem_size > 500000 else PrecisionLevel.DOUBLE
        return PrecisionLevel.SINGLE

    def get_precision_for_level(self, level, max_levels):                    # core/precision.py:337-357
        if not self.adaptive or self.current_precision != PrecisionLevel.MIXED:
            return self.current_precision
        return PrecisionLevel.SINGLE if level >= max_levels // 2 else PrecisionLevel.DOUBLE

    def get_statistics(self):                                                # core/precision.py:359-385
        total_ops = sum(s["operations"] for s in self.precision_stats.values())
        total_time = sum(s["time"] for s in self.precision_stats.values())
        return {
            "current_precision": self.current_precision.value,
            "precision_history": [p.value for p in self.precision_history],
            "total_operations": total_ops,
            "total_time": total_time,
            "precision_breakdown": {
                lvl.value: {"operations": self.precision_stats[lvl]["operations"],
                            "time": self.precision_stats[lvl]["time"],
                            "percentage": (self.precision_stats[lvl]["operations"] / total_ops * 100
                                           if total_ops > 0 else 0)}
                for lvl in PrecisionLevel if lvl != PrecisionLevel.MIXED},
        }

    def record_operation(self, precision, time_taken):
        if precision in self.precision_stats:
            self.precision_stats[precision]["operations"] += 1
            self.precision_stats[precision]["time"] += time_taken

    def reset_statistics(self):
        for lvl in self.precision_stats:
            self.precision_stats[lvl]["operations"] = 0
            self.precision_stats[lvl]["time"] = 0.0
        self.precision_history = [self.current_precision]

    def __repr__(self):
        return (f"PrecisionManager(default={self.default_precision.value}, "
                f"current={self.current_precision.value}, adaptive={self.adaptive}, "
                f"threshold={self.convergence_threshold})")
