#!/usr/bin/env python3
"""V(2,2) weighted-Jacobi cycle times of square and 2:1 hierarchies (fp64), with the direct coarsest solve (default) and the
reference's iteration: 2:1 grids end in 9 x 5 / 5 x 9 and run the LDS tail -- what the replicated coarse engine of a 2 x 1 or
4 x 2 decomposition solves every cycle.    python3 tools/rect_probe.py"""
import sys, numpy as np
sys.path.insert(0, ".")
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
for nx, ny in ((1025, 1025), (1025, 513), (513, 1025), (2049, 1025), (513, 513), (513, 257)):
    for direct in (None, False):
        e = mg.MultigridEngine(nx, ny, domain=(0.0, (nx - 1) / (ny - 1) if nx >= ny else 1.0, 0.0, 1.0 if nx >= ny else (ny - 1) / (nx - 1)),
                               max_levels=mg.default_max_levels(nx, ny), smoother=_lib.MG_JACOBI, omega=0.8, coarse_direct=direct)
        x = np.linspace(0, 1, nx); y = np.linspace(0, 1, ny)
        e.set_rhs(np.sin(np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]); e.set_solution(None); e.iterate(0.0, 3)
        e.set_solution(None); r = e.iterate(0.0, 20)
        print(f"{nx}x{ny} direct={direct}: {r['solve_seconds'] / 20 * 1e6:7.1f} us/cycle, levels {e.num_levels}, coarsest {e.shapes[-1]}, sweeps {r.get('last_coarse_sweeps')}", flush=True)
        e.close()
