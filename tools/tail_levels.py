#!/usr/bin/env python3
"""Duration of the single-workgroup LDS tail (levels <= 65^2 + coarsest solve) by its top level: whole V / W cycles of
n = 9 .. 65 grids are one coarse_tail_kernel launch each.

    python3 tools/tail_levels.py [reps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for cyc in ("V", "W"):
    for n in (9, 17, 33, 65):
        x = np.linspace(0, 1, n)
        rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
        for prec, name in ((_lib.MG_PREC_DOUBLE, "f64"), (_lib.MG_PREC_SINGLE_MANAGED, "f32+f64 coarsest")):
            eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle=cyc, precision=prec)
            eng.set_rhs(rhs)
            eng.set_solution(None)
            eng.cycle(1)
            us = eng.time_op("cycle", 0, np.float64 if prec == _lib.MG_PREC_DOUBLE else np.float32, reps) * 1e3
            print(f"{cyc}-cycle tail from {n:3d}^2 ({eng.num_levels} levels) {name:18s}: {us:7.2f} us / launch")
            eng.close()
