#!/usr/bin/env python3
"""A few W(2,2) red-black GS cycles at n^2 (fp64), for rocprofv3 --kernel-trace --stats runs: where a W-cycle's time goes
(level l is visited 2^l times; the LDS tail -- 65^2 and below -- 2^(L-6) times per cycle).

    python3 tools/wcycle_probe.py [n] [cycles] [coarse_direct 0|1|auto (default: the engine's own choice, direct in W-cycles)]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 3
direct = {"0": False, "1": True}.get(sys.argv[3] if len(sys.argv) > 3 else "auto", "auto")
x = np.linspace(0, 1, n)
rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle="W", smoother=_lib.MG_RBGS, omega=1.0, coarse_direct=direct)
eng.set_rhs(rhs)
eng.set_solution(None)
eng.iterate(0.0, 1)
eng.set_solution(None)
r = eng.iterate(0.0, cycles)
print(f"{n}^2 W(2,2) red-black GS fp64, coarse_direct={direct}: {r['solve_seconds'] / cycles * 1e3:.3f} ms/cycle, "
      f"||r|| {r['residual_history'][0]:.2e} -> {r['residual_history'][-1]:.2e}")
eng.close()
