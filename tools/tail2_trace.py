"""Where a visit of the register-resident coarse tail spends its time: s_memtime stamps inside tail2_kernel
(csrc/mg_tail_kernels.hpp, T2_STAMP) of the last launch of a few W(2,2) / V(2,2) cycles.  Needs a measurement build:

    MGHIP_EXTRA_FLAGS="-DMG_EXPERIMENTS -DMG_EXP_TAIL_TRACE=1" MGHIP_LIBRARY_OUT=/tmp/libmghip_trace.so python3 -m mixed_precision_multigrid_solvers_for_pdes_amd._build
    MGHIP_LIBRARY=/tmp/libmghip_trace.so python3 tools/tail2_trace.py [n] [cycle] [smoother] [direct]
"""
import ctypes as C
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, ".")
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
cyc = sys.argv[2] if len(sys.argv) > 2 else "W"
sm = {"rbgs": _lib.MG_RBGS, "jacobi": _lib.MG_JACOBI}[sys.argv[3] if len(sys.argv) > 3 else "rbgs"]
direct = {"1": True, "0": False}.get(sys.argv[4] if len(sys.argv) > 4 else "", None)
x = np.linspace(0, 1, n)
f = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
e = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle=cyc, smoother=sm, omega=1.0 if sm == _lib.MG_RBGS else 0.8,
                       coarse_direct=direct)
e.set_rhs(f)
e.set_solution(None)
e.iterate(0.0, 2)
e.synchronize()
lib = _lib.load()
buf = (C.c_longlong * 2050)()
lib.mg_exp_tail2_trace.restype = C.c_int
assert lib.mg_exp_tail2_trace(buf, 2050) == 0
cnt = int(buf[0])
ev = [(int(buf[2 + 2 * k]), int(buf[3 + 2 * k])) for k in range(cnt)]
print(f"{n}^2 {cyc}-cycle, {cnt} stamps, whole launch {(ev[-1][1] - ev[0][1]) / 100.0:.2f} us (wall_clock64: 100 MHz)")
names = {0: "enter", 1: "pre sweeps", 2: "residual + restrict (+sync)", 3: "levels below (+sync)", 4: "interpolation", 5: "post sweeps"}
tot, num = defaultdict(float), defaultdict(int)
for (i0, t0), (i1, t1) in zip(ev[:-1], ev[1:]):
    if i1 >= 90:
        key = {91: "load top level", 92: "store top level"}.get(i1, str(i1))
    else:
        lvl, ph = divmod(i1, 10)
        if ph == 0:
            key = f"level {lvl}: (between visits)"
        elif ph == 3 and i0 // 10 == lvl:
            key = f"level {lvl}: 5x5 solves (+sync)" if i0 % 10 == 2 else f"level {lvl}: {names[ph]}"
        else:
            key = f"level {lvl}: {names[ph]}"
    tot[key] += (t1 - t0) / 100.0
    num[key] += 1
for k in sorted(tot):
    print(f"  {k:42s} {num[k]:4d} x {tot[k] / num[k]:7.3f} us = {tot[k]:8.2f} us")
