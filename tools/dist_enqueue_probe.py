#!/usr/bin/env python3
"""Decomposed cycle with virtual ranks: wall time per cycle against the time the host spends enqueueing it (the cycle() calls of
the replayed plans).  At 4097^2 per rank the two coincide because the GPU is the limit and the queue pushes back; at 2049^2 per
rank (0.15 ms per rank-cycle) the host's ~3-5 us per operation is what is left.    python3 tools/dist_enqueue_probe.py px py n"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from mixed_precision_multigrid_solvers_for_pdes_amd import distributed as D
px, py, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
NX, NY = px * (n - 1) + 1, py * (n - 1) + 1
dom = (0.0, float(px), 0.0, float(py))
ops = D.HipOps(np.float64, torch.device("cuda", 0))
sv = D.DistributedMultigrid(NX, NY, px, py, range(px * py), ops, None, domain=dom, smoother="jacobi", omega=0.8, native=True)
sv.set_problem(lambda b: D.sine_rhs_block(b, dom))
for _ in range(5):
    sv.cycle(0); sv.residual_norm()
torch.cuda.synchronize()
K = 20
t_enq = 0.0
t0 = time.perf_counter()
for _ in range(K):
    a = time.perf_counter(); sv.cycle(0); t_enq += time.perf_counter() - a
    sv.residual_norm()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{px}x{py} of {n}^2: wall {wall / K * 1e3:.3f} ms per cycle, of which enqueueing (cycle() calls) {t_enq / K * 1e3:.3f} ms")
# how long would the GPU take alone?  enqueue 20 cycles without looking at any norm
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    sv.cycle(0)
t1 = time.perf_counter() - t0
torch.cuda.synchronize(); wall2 = time.perf_counter() - t0
print(f"   without collecting norms in between: enqueue {t1 / K * 1e3:.3f} ms per cycle, wall {wall2 / K * 1e3:.3f} ms per cycle")
sv.close()
