#!/usr/bin/env python3
"""Host-side cost of the distributed driver: all ranks of a px x py decomposition as virtual ranks on ONE GPU.
With small blocks the GPU work is negligible, so ms/cycle ~ Python + launch overhead of the orchestration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixed_precision_multigrid_solvers_for_pdes_amd import distributed as D
px, py, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
NX, NY = px * (n - 1) + 1, py * (n - 1) + 1
dom = (0.0, float(px), 0.0, float(py))
ops = D.HipOps(np.float32, torch.device("cuda", 0), managed_single=True)
s = D.DistributedMultigrid(NX, NY, px, py, range(px * py), ops, None, domain=dom, smoother="jacobi", omega=0.8)
s.set_problem(lambda b: D.sine_rhs_block(b, dom))
for _ in range(2):
    s.cycle(0); s.residual_norm()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 5
for _ in range(K):
    s.cycle(0); s.residual_norm()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"{px}x{py} virtual ranks, {n}^2 each ({NX}x{NY}), Ld={s.Ld}: {dt*1e3:.2f} ms/cycle -> {dt*1e3/(px*py):.2f} ms per rank-cycle")
s.close()
