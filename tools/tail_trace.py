#!/usr/bin/env python3
"""Where the coarse tail spends its time: s_memtime stamps inside coarse_tail_kernel (timing-only build
-DMG_EXP_TAIL_TRACE=1, MGHIP_LIBRARY pointing at it) after one V-cycle of the 4097^2 bench problem.

    MGHIP_LIBRARY=.../exp_trace.so python3 tools/tail_trace.py [n]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
x = np.linspace(0, 1, n)
rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), precision=_lib.MG_PREC_DOUBLE)
eng.set_rhs(rhs)
eng.set_solution(None)
lib = _lib.load()
names = ["prologue (zero pool, load top)"] + ["down 65", "down 33", "down 17", "down 9", "solve 5", "up 9", "up 17", "up 33", "up 65"] + ["store top"]
for cyc in range(6):
    eng.cycle(1)
    eng.residual_norm()
    buf = (C.c_longlong * 64)()
    lib.mg_exp_tail_trace(buf)
    t = np.array(buf[:12], dtype=np.int64)
    d = np.diff(t)
    print(f"cycle {cyc + 1}: total {int(t[-1] - t[0])} ticks; " + ", ".join(f"{nm} {int(v)}" for nm, v in zip(names, d)))
eng.close()
