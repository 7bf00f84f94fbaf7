"""Shifted operator -Laplace_h + sigma (mg_set_shift) and the heat-equation time stepper on top of it.

Pinning: sigma = 0 is the reference's operator bit for bit; for sigma > 0 the reference has no multigrid (its
heat_equation.py relaxes the same linear system with 100 Gauss-Seidel sweeps), so the kernels are checked against the
oracle's shifted functions (mg_oracle, shift=...) bit for bit, whole solves against MGOracle(shift=...), and the time
stepper against (i) oracle.heat_oracle -- itself pinned to the reference's outputs by tests/test_heat_golden.py --
run to convergence, (ii) the reference's own outputs in tests/golden/heat.npz within the iteration error its fixed
100 sweeps leave, (iii) the exact amplification factor of a discrete eigenmode at 1025^2."""
import os
import sys

import numpy as np
import pytest

import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
from mixed_precision_multigrid_solvers_for_pdes_amd import heat_equation as H
from oracle import mg_oracle as O
from oracle.heat_oracle import HeatOracle

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from heat_inputs import heat_cases, heat_config                                   # noqa: E402

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("shape,domain", [((33, 33), (0.0, 1.0, 0.0, 1.0)), ((65, 129), (0.0, 1.0, 0.0, 1.0)),
                                          ((257, 257), (0.0, 1.0, 0.0, 1.0)), ((21, 13), (0.0, 1.5, -0.2, 0.5))])
@pytest.mark.parametrize("sigma", [160.0, 0.37, 4096.0])
def test_shifted_operators_equal_oracle(shape, domain, dt, sigma):
    nx, ny = shape
    rng = np.random.default_rng(nx * ny)
    u = rng.standard_normal(shape).astype(dt); f = rng.standard_normal(shape).astype(dt)
    grid = mg.Grid(nx, ny, domain=domain, dtype=dt)
    hx, hy = O.grid_spacing(nx, ny, domain)
    op = mg.HelmholtzOperator(sigma)
    exact = domain == (0.0, 1.0, 0.0, 1.0)            # dyadic spacings: reciprocals exact -> bit equality
    def same(a, b):
        if exact:
            np.testing.assert_array_equal(a, b)
        else:
            assert rel(a, b) < (1e-13 if dt == np.float64 else 1e-5)
    same(op.residual(grid, u, f), O.residual(u, f, hx, hy, -1.0, sigma))
    same(op.apply(grid, u), O.apply_laplacian(u, hx, hy, -1.0, sigma))
    same(mg.JacobiSmoother(relaxation_parameter=0.8).smooth(grid, op, u, f, 3), O.jacobi(u, f, hx, hy, 0.8, 3, shift=sigma))
    same(mg.GaussSeidelSmoother(red_black=True, relaxation_parameter=1.15).smooth(grid, op, u, f, 2),
         O.rbgs(u, f, hx, hy, 1.15, 2, shift=sigma))
    # sigma = 0 is LaplacianOperator bit for bit
    zero, lap = mg.HelmholtzOperator(0.0), mg.LaplacianOperator(coefficient=-1.0)
    np.testing.assert_array_equal(zero.residual(grid, u, f), lap.residual(grid, u, f))
    np.testing.assert_array_equal(mg.JacobiSmoother().smooth(grid, zero, u, f, 2), mg.JacobiSmoother().smooth(grid, lap, u, f, 2))
    with pytest.raises(ValueError):
        mg.HelmholtzOperator(-1.0)


@pytest.mark.parametrize("fused,tail", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("n,cyc,kind,omega,prec,sigma", [
    (65, "V", "jacobi", 0.8, "double", 320.0), (129, "V", "rbgs", 1.0, "double", 50.0), (65, "W", "rbgs", 1.0, "double", 1e4),
    (33, "F", "jacobi", 0.8, "double", 3.0), (129, "V", "jacobi", 0.8, "mixed", 640.0), (129, "W", "jacobi", 2.0 / 3.0, "single", 77.0)])
def test_shifted_cycles_equal_oracle(n, cyc, kind, omega, prec, sigma, fused, tail):
    rng = np.random.default_rng(n)
    dtype = np.float32 if prec == "single" else np.float64
    rhs = (O.sine_rhs(n, n) + 0.05 * rng.standard_normal((n, n))).astype(dtype)
    u0 = (0.1 * rng.standard_normal((n, n))).astype(dtype)           # non-zero Dirichlet ring + interior guess
    levels = mg.default_max_levels(n, n)
    ref = O.MGOracle(n, n, dtype=dtype, max_levels=levels, cycle=cyc, smoother=kind, omega=omega, shift=sigma)
    pm = O.OraclePrecision("mixed") if prec == "mixed" else None
    u_ref, info = ref.solve(rhs, u0, tol=0.0, max_iterations=3, pm=pm)
    code = {"double": _lib.MG_PREC_DOUBLE, "mixed": _lib.MG_PREC_MIXED_LEVELS, "single": _lib.MG_PREC_SINGLE}[prec]
    eng = mg.MultigridEngine(n, n, max_levels=levels, cycle=cyc, smoother=_lib.MG_JACOBI if kind == "jacobi" else _lib.MG_RBGS,
                             omega=omega, precision=code, fused=fused, tail=tail)
    eng.set_shift(sigma)
    u, r = eng.solve(rhs, u0, tol=0.0, max_iterations=3)
    tol_u = 1e-12 if prec == "double" else 1e-5
    assert rel(u, u_ref) <= tol_u
    np.testing.assert_allclose(r["residual_history"], info["residual_history"], rtol=1e-9 if prec == "double" else 2e-3)
    # back to sigma = 0: the unshifted engine, bit for bit
    eng.set_shift(0.0)
    ua, _ = eng.solve(rhs, u0, tol=0.0, max_iterations=2)
    eng2 = mg.MultigridEngine(n, n, max_levels=levels, cycle=cyc, smoother=_lib.MG_JACOBI if kind == "jacobi" else _lib.MG_RBGS,
                              omega=omega, precision=code, fused=fused, tail=tail)
    ub, _ = eng2.solve(rhs, u0, tol=0.0, max_iterations=2)
    np.testing.assert_array_equal(ua, ub)
    eng.close(); eng2.close()


def test_shift_rejects_bad_values():
    eng = mg.MultigridEngine(33, 33, max_levels=4)
    with pytest.raises(ValueError):
        eng.set_shift(-1.0)
    with pytest.raises(ValueError):
        eng.set_shift(float("nan"))
    eng.close()


def test_shift_with_variable_coefficient_equals_oracle():
    n, sigma = 65, 90.0
    x = np.linspace(0, 1, n); X, Y = np.meshgrid(x, x, indexing="ij")
    a = 1.0 + 0.5 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y)
    rhs = O.sine_rhs(n, n); rhs[0, :] = rhs[-1, :] = rhs[:, 0] = rhs[:, -1] = 0.0
    levels = mg.default_max_levels(n, n)
    ref = O.VarMGOracle(a, max_levels=levels, cycle="V", smoother="rbgs", omega=1.0, shift=sigma)
    ref.rhs[0] = rhs.copy()
    u_ref = np.zeros_like(rhs); h_ref = []
    for _ in range(3):
        u_ref = ref.cycle_once(u_ref, 0); h_ref.append(ref.residual_norm(u_ref, rhs, 0))
    eng = mg.MultigridEngine(n, n, max_levels=levels, cycle="V", smoother=_lib.MG_RBGS, omega=1.0)
    eng.set_coefficient(a); eng.set_shift(sigma)
    u, r = eng.solve(rhs, tol=0.0, max_iterations=3)
    assert rel(u, u_ref) <= 1e-13
    np.testing.assert_allclose(r["residual_history"], h_ref, rtol=1e-9)
    eng.close()


def test_solver_class_with_helmholtz_operator():
    """MultigridSolver.solve(grid, HelmholtzOperator, ...): the shift is read per solve (time steppers change it)."""
    n = 129
    grid = mg.Grid(n, n)
    ue = np.sin(np.pi * grid.X) * np.sin(2 * np.pi * grid.Y)
    s = mg.MultigridSolver(max_levels=mg.default_max_levels(n, n), max_iterations=30, tolerance=1e-9)
    s.setup(grid, mg.HelmholtzOperator(10.0), mg.RestrictionOperator(), mg.ProlongationOperator(), smoother=mg.WeightedJacobiSmoother())
    errs = []
    for sigma in (10.0, 1000.0):
        op = mg.HelmholtzOperator(sigma)
        f = (5 * np.pi**2 + sigma) * ue
        f[0, :] = f[-1, :] = f[:, 0] = f[:, -1] = 0.0
        u, info = s.solve(grid, op, f)
        assert info["converged"]
        errs.append(np.max(np.abs(u - ue)))
    s.cleanup()
    assert max(errs) < 2e-3          # O(h^2) discretisation error of the (1,2) mode at h = 1/128


# ------------------------------------------------------------------ heat equation ---------------
@pytest.fixture(scope="module")
def heat_golden():
    return np.load(os.path.join(HERE, "golden", "heat.npz"))


@pytest.mark.parametrize("name", sorted(heat_cases()))
def test_heat_steps_equal_converged_oracle_and_reference(heat_golden, name):
    n, alpha, scheme, _, steps, bc_kind, with_source = heat_cases()[name]
    dt = float(heat_golden[f"{name}__dt"])
    cfg = heat_config(H, alpha, bc_kind, with_source)
    hs = H.HeatEquationSolver(cfg, mg.Grid(n, n))
    conv = HeatOracle(cfg, n, n, sweeps=20000)        # the same relaxation run to convergence (stops at its own 1e-10 test? no: never) 
    u = hs.set_initial_condition(heat_golden[f"{name}__u0"]).copy()
    uo = conv.set_initial_condition(heat_golden[f"{name}__u0"]).copy()
    sch = H.TimeSteppingScheme(scheme)
    worst_ref = 0.0
    for k in range(steps):
        # one step from the SAME state (the reference's), so iteration errors do not accumulate in the comparison
        prev = heat_golden[f"{name}__u{k}"]
        hs.current_time = conv.t = k * dt
        u = hs._single_time_step(prev.copy(), dt, sch)
        uo = conv.step(prev.copy(), dt, scheme)
        assert rel(u, uo) < 1e-9, (name, k)
        worst_ref = max(worst_ref, rel(u, heat_golden[f"{name}__u{k + 1}"]))
    # the reference itself: exact for the explicit scheme, within its Gauss-Seidel iteration error otherwise
    assert worst_ref < (1e-12 if scheme == "explicit_euler" else 5e-3), worst_ref
    if scheme != "explicit_euler":
        assert all(c <= 20 for _, c, _ in hs.helmholtz_stats)


def test_heat_adaptive_run_matches_reference(heat_golden):
    cfg = heat_config(H, 1.0, "zero", False)
    hs = H.HeatEquationSolver(cfg, mg.Grid(17, 17))
    hs.set_initial_condition()
    res = hs.solve_time_dependent(0.02, 0.004, H.TimeSteppingScheme.CRANK_NICOLSON, adaptive=True, error_tolerance=2e-3)
    assert res["total_steps"] == int(heat_golden["adaptive17__steps"])
    np.testing.assert_allclose(res["dt_history"], heat_golden["adaptive17__dts"], rtol=1e-6)
    assert abs(res["final_time"] - 0.02) < 1e-12
    assert rel(res["final_solution"], heat_golden["adaptive17__final"]) < 1e-6
    assert set(res) == {"solution_history", "time_history", "dt_history", "final_solution", "final_time", "total_steps",
                        "solve_time", "scheme", "adaptive"}
    with pytest.raises(ValueError):
        H.HeatEquationSolver(cfg, mg.Grid(17, 17)).solve_time_dependent(0.1)


@pytest.mark.parametrize("scheme", ["implicit_euler", "crank_nicolson"])
def test_heat_eigenmode_amplification_1025(scheme):
    """sin(pi x) sin(2 pi y) is an eigenvector of the 5-point Laplacian: one implicit step multiplies it by
    1/(1 + dt a mu) (implicit Euler) or (1 - dt a mu/2)/(1 + dt a mu/2) (Crank-Nicolson), mu the discrete eigenvalue."""
    n, alpha, dt = 1025, 0.7, 2e-3
    g = mg.Grid(n, n)
    h = g.hx
    mode = np.sin(np.pi * g.X) * np.sin(2 * np.pi * g.Y)
    mu = (4 / h**2) * (np.sin(np.pi * h / 2) ** 2 + np.sin(2 * np.pi * h / 2) ** 2)
    hs = H.HeatEquationSolver(H.HeatEquationConfig(thermal_diffusivity=alpha), g)
    hs.set_initial_condition(mode)
    u = hs._single_time_step(mode.copy(), dt, H.TimeSteppingScheme(scheme))
    z = dt * alpha * mu
    factor = 1 / (1 + z) if scheme == "implicit_euler" else (1 - z / 2) / (1 + z / 2)
    assert rel(u, factor * mode) < 1e-9
    assert hs.helmholtz_stats[-1][1] <= 12
