#!/usr/bin/env python3
"""Time the fused legs level by level (level 0 up leg includes the norm stage, deeper ones do not).

    python3 tools/leg_levels.py [n] [reps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = np.linspace(0, 1, n)
rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
FUSED = int(os.environ.get("MG_FUSED", "1"))      # 1 LDS-tiled, 2 register-blocked on large levels, 3 register-blocked everywhere
eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), precision=_lib.MG_PREC_ADAPTIVE, fused=FUSED)
eng.set_rhs(rhs)
eng.set_solution(None)
eng.cycle(1)
for level in range(0, min(7, eng.num_levels - 1)):
    m = eng.shapes[level][0]
    for dt, w in ((np.float32, 4), (np.float64, 8)):
        row = [f"level {level} {m:5d}^2 {np.dtype(dt).name}:"]
        for op in ("sweeps2", "down_leg", "up_leg"):
            us = eng.time_op(op, level, dt, reps) * 1e3
            row.append(f"{op} {us:8.2f} us ({3.25 * m * m * w / us / 1e6:5.2f} TB/s of 3.25w)")
        print("  ".join(row))
eng.close()
