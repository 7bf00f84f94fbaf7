"""Variable-coefficient operator -div(a grad u) (BASELINE config 5).  The reference has no implementation:
PARITY UNPINNED.  What pins it: (i) a == 1 reproduces the constant-coefficient kernels bit for bit, (ii) the GPU
path equals our own NumPy restatement (oracle VarMGOracle) bit for bit, (iii) second-order convergence on a
manufactured solution."""
import numpy as np
import pytest

import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
from oracle import mg_oracle as O

pytestmark = pytest.mark.gpu


def _coef(nx, ny, dtype=np.float64):
    x = np.linspace(0, 1, nx); y = np.linspace(0, 1, ny)
    X, Y = np.meshgrid(x, y, indexing="ij")
    return (1.0 + 0.5 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y)).astype(dtype), X, Y


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(33, 33), (65, 129), (257, 257)])
def test_var_operators_equal_oracle_and_reduce_to_constant(shape, dt):
    nx, ny = shape
    rng = np.random.default_rng(nx + ny)
    u = rng.standard_normal(shape).astype(dt); f = rng.standard_normal(shape).astype(dt)
    a, _, _ = _coef(nx, ny, dt)
    grid = mg.Grid(nx, ny, dtype=dt)
    hx, hy = O.grid_spacing(nx, ny)
    op = mg.DiffusionOperator(a)
    np.testing.assert_array_equal(op.residual(grid, u, f), O.var_residual(u, f, a, hx, hy))
    np.testing.assert_array_equal(mg.JacobiSmoother(relaxation_parameter=0.8).smooth(grid, op, u, f, 3), O.var_jacobi(u, f, a, hx, hy, 0.8, 3))
    np.testing.assert_array_equal(mg.GaussSeidelSmoother(red_black=True, relaxation_parameter=1.15).smooth(grid, op, u, f, 2),
                                  O.var_rbgs(u, f, a, hx, hy, 1.15, 2))
    # a == 1: the constant-coefficient kernels -- operator and residual bit for bit; the sweeps bit for bit where the diagonal
    # is a power of two (square dyadic cells: its reciprocal is exact), else to the one rounding that separates
    # x * fl(1 / D) (variable-coefficient sweeps multiply by the stored reciprocal diagonal) from x / D
    one = mg.DiffusionOperator(np.ones(shape, dtype=dt))
    lap = mg.LaplacianOperator(coefficient=-1.0)
    np.testing.assert_array_equal(one.residual(grid, u, f), lap.residual(grid, u, f))
    np.testing.assert_array_equal(one.apply(grid, u), lap.apply(grid, u))
    same = np.testing.assert_array_equal if nx == ny else (lambda x, y: np.testing.assert_allclose(x, y, rtol=0, atol=8 * np.finfo(dt).eps * np.max(np.abs(y))))
    same(mg.JacobiSmoother(relaxation_parameter=0.8).smooth(grid, one, u, f, 2),
         mg.JacobiSmoother(relaxation_parameter=0.8).smooth(grid, lap, u, f, 2))
    same(mg.GaussSeidelSmoother(red_black=True).smooth(grid, one, u, f, 2),
         mg.GaussSeidelSmoother(red_black=True).smooth(grid, lap, u, f, 2))
    with pytest.raises(ValueError):
        mg.DiffusionOperator(-np.ones(shape)).residual(grid, u, f)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("n,cyc,kind,omega,prec", [(65, "V", "jacobi", 0.8, "double"), (129, "V", "rbgs", 1.0, "double"), (65, "W", "rbgs", 1.0, "double"),
                                                   (33, "F", "jacobi", 0.8, "double"), (129, "W", "rbgs", 1.0, "mixed"), (257, "V", "jacobi", 0.8, "mixed"),
                                                   (1025, "V", "rbgs", 1.0, "mixed"), (1025, "V", "jacobi", 0.8, "double"), (129, "V", "rbgs", 1.15, "single")])
def test_var_cycles_equal_oracle(n, cyc, kind, omega, prec, fused):
    """The variable-coefficient cycle (fused legs + LDS tail, and one launch per operator) against the NumPy restatement
    (oracle.VarMGOracle, our own design: parity unpinned) incl. per-level mixed precision at >= 1025^2."""
    a, X, Y = _coef(n, n)
    rng = np.random.default_rng(n)
    rhs = O.sine_rhs(n, n) + 0.05 * rng.standard_normal((n, n))
    rhs[0, :] = rhs[-1, :] = rhs[:, 0] = rhs[:, -1] = 0.0
    levels = mg.default_max_levels(n, n)
    dt = np.float32 if prec == "single" else np.float64
    ref = O.VarMGOracle(a, (0.0, 1.0, 0.0, 1.0), dt, max_levels=levels, cycle=cyc, smoother=kind, omega=omega,
                        coarse_maxit=1000 if prec != "single" else 50)
    pm = O.OraclePrecision("mixed") if prec == "mixed" else None
    ref.rhs[0] = rhs.astype(dt)
    u_ref = np.zeros_like(ref.rhs[0]); h_ref = []
    ncyc = 2 if n >= 1025 else 3
    for _ in range(ncyc):
        u_ref = ref.cycle_once(u_ref, 0, pm)
        h_ref.append(ref.residual_norm(u_ref.astype(dt), rhs.astype(dt), 0))
    code = {"double": _lib.MG_PREC_DOUBLE, "mixed": _lib.MG_PREC_MIXED_LEVELS, "single": _lib.MG_PREC_SINGLE}[prec]
    eng = mg.MultigridEngine(n, n, max_levels=levels, cycle=cyc, smoother=_lib.MG_JACOBI if kind == "jacobi" else _lib.MG_RBGS, omega=omega,
                             precision=code, fused=fused, coarse_maxit=1000 if prec != "single" else 50)
    eng.set_coefficient(a)
    u, r = eng.solve(rhs.astype(dt), tol=0.0, max_iterations=ncyc)
    tol = 1e-13 if prec != "single" else 2e-6
    assert np.max(np.abs(u - u_ref)) <= tol * np.max(np.abs(u_ref))          # coarsest stop test may differ by a sweep
    np.testing.assert_allclose(r["residual_history"], h_ref, rtol=1e-9 if prec != "single" else 1e-4)
    if prec != "double" or n >= 1025:
        eng.close()
        return
    # a == 1 through the variable-coefficient path == the constant-coefficient engine (one launch per operator)
    eng.set_coefficient(np.ones((n, n)))
    u1, _ = eng.solve(rhs, tol=0.0, max_iterations=3)
    eng.set_coefficient(None)
    u2, _ = eng.solve(rhs, tol=0.0, max_iterations=3)
    assert np.max(np.abs(u1 - u2)) <= 1e-13 * np.max(np.abs(u2))
    eng.close()


@pytest.mark.parametrize("strategy", ["double", "mixed"])
def test_var_manufactured_solution_second_order(strategy):
    """u = sin(pi x) sin(pi y), a = 1 + 0.5 sin(2 pi x) cos(2 pi y): max error ~ h^2."""
    errs = []
    for n in (65, 129, 257):
        a, X, Y = _coef(n, n)
        ue = np.sin(np.pi * X) * np.sin(np.pi * Y)
        ax = np.pi * np.cos(2 * np.pi * X) * np.cos(2 * np.pi * Y); ay = -np.pi * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y)
        ux = np.pi * np.cos(np.pi * X) * np.sin(np.pi * Y); uy = np.pi * np.sin(np.pi * X) * np.cos(np.pi * Y)
        f = -(ax * ux + ay * uy + a * (-2 * np.pi**2 * ue))
        f[0, :] = f[-1, :] = f[:, 0] = f[:, -1] = 0.0
        grid = mg.Grid(n, n)
        op = mg.DiffusionOperator(lambda x, y: 1.0 + 0.5 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y))
        s = mg.MultigridSolver(max_levels=mg.default_max_levels(n, n), max_iterations=40, tolerance=1e-9, cycle_type="W" if n == 65 else "V")
        s.setup(grid, op, mg.RestrictionOperator(), mg.ProlongationOperator(), smoother=mg.GaussSeidelSmoother(red_black=True))
        pm = mg.PrecisionManager(default_precision="mixed") if strategy == "mixed" else None
        u, info = s.solve(grid, op, f, None, pm)
        s.cleanup()
        assert info["converged"], info["residual_history"][-3:]
        errs.append(np.max(np.abs(u - ue)))
    assert 1.9 < np.log2(errs[0] / errs[1]) < 2.1 and 1.9 < np.log2(errs[1] / errs[2]) < 2.1, errs


@pytest.mark.parametrize("sm,omega", [("jacobi", 0.8), ("rbgs", 1.0), ("rbgs", 1.15)])
@pytest.mark.parametrize("prec", ["double", "single", "mixed", "adaptive"])
@pytest.mark.parametrize("n,cyc,pre,post", [(257, "V", 2, 2), (129, "W", 2, 2), (513, "V", 1, 1), (129, "V", 3, 0), (65, "F", 0, 4),
                                            ((97, 193), "V", 2, 1), (1025, "V", 2, 2)])
def test_var_fused_legs_equal_one_launch_per_operator(prec, n, cyc, pre, post, sm, omega):
    """Variable-coefficient fused down / up legs (coefficient tile staged with the iterate, face means in registers) and
    the variable-coefficient LDS tail must reproduce the operator-by-operator cycle bit for bit."""
    nx, ny = (n, n) if isinstance(n, int) else n
    code = {"double": _lib.MG_PREC_DOUBLE, "single": _lib.MG_PREC_SINGLE, "mixed": _lib.MG_PREC_MIXED_LEVELS,
            "adaptive": _lib.MG_PREC_ADAPTIVE}[prec]
    rng = np.random.default_rng(nx + ny + pre)
    rhs = O.sine_rhs(nx, ny) + 0.05 * rng.standard_normal((nx, ny))
    u0 = rng.standard_normal((nx, ny))
    a = np.exp(0.6 * rng.standard_normal((nx, ny)))                 # rough, positive: nothing cancels by symmetry
    res = []
    for fused, tail in ((1, True), (1, False), (0, False), (3, True)):      # 3: the register-blocked legs on every level
        eng = mg.MultigridEngine(nx, ny, max_levels=mg.default_max_levels(nx, ny), cycle=cyc, pre=pre, post=post,
                                 smoother=_lib.MG_JACOBI if sm == "jacobi" else _lib.MG_RBGS, omega=omega, precision=code,
                                 switch_threshold=1e-3, coarse_maxit=60, fused=fused, tail=tail, speculate=tail)
        eng.set_coefficient(a)
        u, r = eng.solve(rhs, u0, tol=1e-30, max_iterations=5)
        eng.close()
        res.append((u, r))
    (ut, rt), (uf, rf), (uu, ru), (ub, rb) = res
    np.testing.assert_array_equal(uf, uu)
    np.testing.assert_array_equal(ut, uu)
    np.testing.assert_array_equal(ub, uu)
    np.testing.assert_allclose(rb["residual_history"], ru["residual_history"], rtol=1e-11)
    np.testing.assert_allclose(rf["residual_history"], ru["residual_history"], rtol=1e-11)
    np.testing.assert_allclose(rt["residual_history"], ru["residual_history"], rtol=1e-11)
    assert rf["precision_codes"] == ru["precision_codes"] == rt["precision_codes"]


def test_var_fused_with_helmholtz_shift_and_rectangular_cells():
    """-div(a grad u) + sigma u on a non-dyadic domain (true divisions everywhere): fused == per-operator."""
    nx, ny = 193, 129
    rng = np.random.default_rng(3)
    rhs = rng.standard_normal((nx, ny)); a = 1.0 + rng.random((nx, ny))
    out = []
    for fused in (True, False):
        eng = mg.MultigridEngine(nx, ny, (0.0, 1.5, -0.2, 0.5), max_levels=5, smoother=_lib.MG_RBGS, omega=1.0, cycle="W", fused=fused, coarse_maxit=80)
        eng.set_coefficient(a); eng.set_shift(37.5)
        out.append(eng.solve(rhs, tol=0.0, max_iterations=3))
        eng.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_allclose(out[0][1]["residual_history"], out[1][1]["residual_history"], rtol=1e-11)


def _mms(n):
    a, X, Y = _coef(n, n)
    ue = np.sin(np.pi * X) * np.sin(np.pi * Y)
    ax = np.pi * np.cos(2 * np.pi * X) * np.cos(2 * np.pi * Y); ay = -np.pi * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y)
    ux = np.pi * np.cos(np.pi * X) * np.sin(np.pi * Y); uy = np.pi * np.sin(np.pi * X) * np.cos(np.pi * Y)
    f = -(ax * ux + ay * uy + a * (-2 * np.pi**2 * ue))
    f[0, :] = f[-1, :] = f[:, 0] = f[:, -1] = 0.0
    return a, f, ue


def test_var_manufactured_solution_second_order_at_large_sizes():
    """BASELINE config 5's problem class at 1025^2 / 2049^2 / 4097^2: per-level mixed precision, W(2,2) red-black GS on
    the fused variable-coefficient legs: the discretisation error falls by 4 per refinement."""
    errs = []
    for n in (1025, 2049, 4097):
        a, f, ue = _mms(n)
        eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle="W", smoother=_lib.MG_RBGS, omega=1.0,
                                 precision=_lib.MG_PREC_MIXED_LEVELS)
        eng.set_coefficient(a)
        u, r = eng.solve(f, tol=0.0, max_iterations=8)
        eng.close()
        h = r["residual_history"]
        assert h[3] < 1e-3 * h[0], h
        errs.append(np.max(np.abs(u - ue)))
    assert 1.95 < np.log2(errs[0] / errs[1]) < 2.05 and 1.95 < np.log2(errs[1] / errs[2]) < 2.05, errs


def test_config5_16385_variable_coefficient_mixed_w_rbgs_single_gpu():
    """BASELINE config 5 AS SPECIFIED, on one GPU: -div(a grad u) = f with a = 1 + 0.5 sin(2 pi x) cos(2 pi y) at 16385^2,
    per-level mixed precision, W(2,2) red-black GS, 13 levels.  The fused legs + LDS tail equal the one-launch-per-
    operator cycle bit for bit, and the cycle contracts as on small grids."""
    n = 16385
    x = np.linspace(0.0, 1.0, n)
    a = 1.0 + 0.5 * np.sin(2 * np.pi * x)[:, None] * np.cos(2 * np.pi * x)[None, :]
    sx, cx = np.sin(np.pi * x), np.cos(np.pi * x)
    ue = sx[:, None] * sx[None, :]
    f = -((np.pi * np.cos(2 * np.pi * x))[:, None] * np.cos(2 * np.pi * x)[None, :] * (np.pi * cx[:, None] * sx[None, :])
          + (-np.pi * np.sin(2 * np.pi * x))[:, None] * np.sin(2 * np.pi * x)[None, :] * (np.pi * sx[:, None] * cx[None, :])
          + a * (-2 * np.pi**2 * ue))
    f[0, :] = f[-1, :] = f[:, 0] = f[:, -1] = 0.0
    out = []
    for fused in (True, False):
        eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle="W", smoother=_lib.MG_RBGS, omega=1.0,
                                 precision=_lib.MG_PREC_MIXED_LEVELS, fused=fused)
        eng.set_coefficient(a)
        u, r = eng.solve(f, tol=0.0, max_iterations=2)
        eng.close()
        out.append((u, r["residual_history"]))
    assert np.array_equal(out[0][0], out[1][0])
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-12)
    h = out[0][1]
    assert h[0] < 1e-3 * 20.0 and h[1] < 0.05 * h[0], h           # ||f|| ~ 20: the first W-cycle takes ||r|| down by > 1e3, the next by > 20
    assert np.max(np.abs(out[0][0] - ue)) < 1e-4                   # already close to the exact solution after two cycles
