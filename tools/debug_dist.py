import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib, distributed as D
import dist_helpers as H

def run(px, py, NX, NY, agg, levels=None, dtype=np.float64, kind="jacobi", omega=0.8, cyc="V", ncyc=1, pre=2, post=2):
    rng = np.random.default_rng(NX + NY)
    rhs = rng.standard_normal((NX, NY)).astype(dtype); u0 = rng.standard_normal((NX, NY)).astype(dtype)
    levels = levels or mg.default_max_levels(NX, NY)
    eng = mg.MultigridEngine(NX, NY, max_levels=levels, cycle=cyc, pre=pre, post=post, smoother=0 if kind == "jacobi" else 1, omega=omega,
                             precision=_lib.MG_PREC_SINGLE if dtype == np.float32 else _lib.MG_PREC_DOUBLE)
    eng.set_rhs(rhs); eng.set_solution(u0); eng.cycle(ncyc); u_ref = eng.get_solution(dtype); eng.close()
    ops = D.HipOps(dtype, torch.device("cuda", 0))
    s = D.DistributedMultigrid(NX, NY, px, py, range(px * py), ops, None, max_levels=levels, cycle=cyc, pre=pre, post=post, smoother=kind, omega=omega, agglomerate_at=agg)
    s.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    for _ in range(ncyc): s.cycle(0)
    u = H.assemble(s, NX, NY, dtype); s.close()
    d = np.abs(u - u_ref)
    bad = np.argwhere(d > 0)
    print(f"{px}x{py} {NX}x{NY} agg={agg} L={levels} Ld={s.Ld} pre={pre} post={post}: maxdiff={d.max():.3e} nbad={len(bad)}",
          (f"rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()} argmax={np.unravel_index(d.argmax(), d.shape)}" if len(bad) else ""))

import sys
norm = "norm" in sys.argv
def run2(px, py, NX, NY, agg, ncyc, with_norm, dtype=np.float64):
    rng = np.random.default_rng(NX + NY)
    rhs = rng.standard_normal((NX, NY)).astype(dtype); u0 = rng.standard_normal((NX, NY)).astype(dtype)
    levels = mg.default_max_levels(NX, NY)
    eng = mg.MultigridEngine(NX, NY, max_levels=levels, smoother=0, omega=0.8)
    eng.set_rhs(rhs); eng.set_solution(u0)
    for _ in range(ncyc):
        eng.cycle(1)
        if with_norm: eng.residual_norm()
    u_ref = eng.get_solution(dtype); eng.close()
    eng = mg.MultigridEngine(NX, NY, max_levels=levels, smoother=0, omega=0.8)
    eng.set_rhs(rhs); eng.set_solution(u0); eng.cycle(ncyc); u_ref2 = eng.get_solution(dtype); eng.close()
    ops = D.HipOps(dtype, torch.device("cuda", 0))
    s = D.DistributedMultigrid(NX, NY, px, py, range(px * py), ops, None, max_levels=levels, smoother="jacobi", omega=0.8, agglomerate_at=agg)
    s.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    for _ in range(ncyc):
        s.cycle(0)
        if with_norm: s.residual_norm()
    u = H.assemble(s, NX, NY, dtype); s.close()
    print(f"{px}x{py} {NX}x{NY} ncyc={ncyc} norm={with_norm}: dist-vs-eng {np.abs(u-u_ref).max():.3e}  eng(step)-vs-eng(batch) {np.abs(u_ref-u_ref2).max():.3e}  dist-vs-engbatch {np.abs(u-u_ref2).max():.3e}")
for ncyc in (1, 2, 3):
    for wn in (False, True):
        run2(4, 2, 1025, 513, 129, ncyc, wn)
run2(2, 2, 513, 513, 129, 2, True)
