"""Pins oracle/mg_oracle.py against the golden vectors the REFERENCE produced
(tests/golden/generate_golden.py).  CPU only."""
import re

import numpy as np
import pytest

from oracle import mg_oracle as O

OPS_CASES = ["sq17", "sq33", "rect17x33", "rect33x9", "nondyadic21x13", "min5"]


def _case(g, tag, dt):
    k = f"{tag}_{dt}"
    dom = tuple(float(v) for v in g[f"{k}__domain"])
    u, f = g[f"{k}__u"], g[f"{k}__f"]
    hx, hy = O.grid_spacing(u.shape[0], u.shape[1], dom)
    return k, u, f, hx, hy


@pytest.mark.parametrize("dt", ["float64", "float32"])
@pytest.mark.parametrize("tag", OPS_CASES)
def test_elementwise_ops_bit_exact(golden_ops, tag, dt):
    g = golden_ops
    k, u, f, hx, hy = _case(g, tag, dt)
    eq = np.testing.assert_array_equal
    eq(O.apply_laplacian(u, hx, hy, -1.0), g[f"{k}__apply"])
    eq(O.residual(u, f, hx, hy, -1.0), g[f"{k}__residual"])
    assert O.residual(u, f, hx, hy).dtype == u.dtype
    eq(O.jacobi(u, f, hx, hy, 2.0 / 3.0, 1, "loop"), g[f"{k}__jacobi_w23_nu1"])
    eq(O.jacobi(u, f, hx, hy, 4.0 / 5.0, 2, "loop"), g[f"{k}__jacobi_w08_nu2"])
    eq(O.jacobi(u, f, hx, hy, 0.8, 2, "vectorized"), g[f"{k}__vjacobi_w08_nu2"])
    eq(O.rbgs(u, f, hx, hy, 1.0, 1), g[f"{k}__rbgs_w10_nu1"])
    eq(O.rbgs(u, f, hx, hy, 1.5, 2), g[f"{k}__rbgs_w15_nu2"])
    eq(O.lexgs_sweep(u.copy(), f, hx, hy, 1.0), g[f"{k}__lexgs_w10_nu1"])
    np.testing.assert_allclose(float(O.l2_norm(O.residual(u, f, hx, hy), hx, hy)),
                               float(g[f"{k}__norm"]), rtol=1e-6 if dt == "float32" else 1e-14)
    if f"{k}__restrict_fw" in g.files:
        eq(O.restrict_fw(u), g[f"{k}__restrict_fw"])
        eq(O.prolong_bilinear(g[f"{k}__e"]), g[f"{k}__prolong"])


def test_prolongation_far_edge_quirk(golden_ops):
    """fine[odd i, ny-1] == 0 and fine[nx-1, odd j] == 0 in the reference."""
    p = golden_ops["sq17_float64__prolong"]
    assert np.all(p[1::2, -1] == 0) and np.all(p[-1, 1::2] == 0)
    assert np.any(p[1::2, 0] != 0) and np.any(p[0, 1::2] != 0)


SOLVE_RE = re.compile(r"n(\d+)(?:x(\d+))?_(?:(nondyadic|random)_)?L(\d+)_([VWF])(\d\d)?_([a-z0-9]+)_(float64|float32|mixed|adaptive_ref)__hist")
SMOOTHERS = {"jacobi08": ("jacobi", 4.0 / 5.0, "loop"), "vjacobi08": ("jacobi", 0.8, "vectorized"),
             "jacobi23": ("jacobi", 2.0 / 3.0, "loop"), "rbgs": ("rbgs", 1.0, "loop"),
             "rbgs15": ("rbgs", 1.15, "loop"), "lexgs": ("lexgs", 1.0, "loop")}


def _solve_keys(g):
    return sorted(k for k in g.files if k.endswith("__hist"))


def _run_oracle(g, key):
    m = SOLVE_RE.fullmatch(key)
    assert m, key
    nx = int(m.group(1)); ny = int(m.group(2) or nx)
    special, L, cyc, vv, sm, prec = m.group(3), int(m.group(4)), m.group(5), m.group(6), m.group(7), m.group(8)
    pre, post = (int(vv[0]), int(vv[1])) if vv else (2, 2)
    dom = (0.0, 1.5, -0.2, 0.5) if special == "nondyadic" else (0.0, 1.0, 0.0, 1.0)
    dtype = np.float32 if prec == "float32" else np.float64
    kind, omega, form = SMOOTHERS[sm]
    mg = O.MGOracle(nx, ny, dom, dtype, -1.0, L, cyc, pre, post, kind, omega, form)
    if special == "random":
        rhs, u0, maxit = g["n33_random_rhs"], g["n33_random_u0"], 8
    else:
        rhs, u0 = O.sine_rhs(nx, ny, dom, dtype).astype(dtype), None
        maxit = 12 if prec in ("float32", "adaptive_ref") else 30
    pm = None
    if prec == "mixed":
        pm = O.OraclePrecision("mixed")
    elif prec == "adaptive_ref":
        pm = O.OraclePrecision()
    u, info = mg.solve(rhs, u0, tol=1e-10, max_iterations=maxit, pm=pm)
    return u, info, pm


def test_all_solve_cases_are_parsed(golden_solves):
    keys = _solve_keys(golden_solves)
    assert len(keys) >= 25
    for k in keys:
        assert SOLVE_RE.fullmatch(k), k


@pytest.mark.parametrize("idx", range(28))
def test_solve_histories_and_solutions(golden_solves, idx):
    g = golden_solves
    keys = _solve_keys(g)
    if idx >= len(keys):
        pytest.skip("fewer golden solves than slots")
    key = keys[idx]
    u, info, pm = _run_oracle(g, key)
    ref_hist = g[key]
    hist = np.array(info["residual_history"])
    assert len(hist) == len(ref_hist), key
    f32 = "float32" in key or "adaptive_ref" in key
    # the norm is a pairwise np.sum in both; identical arithmetic => identical to round-off.
    np.testing.assert_allclose(hist, ref_hist, rtol=1e-5 if f32 else 1e-9, atol=1e-13, err_msg=key)
    ukey = key.replace("__hist", "__u")
    if ukey in g.files:
        ref_u = g[ukey]
        assert u.dtype == ref_u.dtype
        scale = np.max(np.abs(ref_u))
        assert np.max(np.abs(u - ref_u)) <= (2e-6 if f32 else 1e-13) * scale, key
    if "adaptive_ref" in key:
        assert pm.history == list(g[key.replace("__hist", "__precisions")])


def test_large_1025_history(golden_large):
    """BASELINE config 2 (1025^2 fp64 V(2,2) Jacobi 0.8, 9 levels): first cycles of the
    reference's own history, and the sampled solution after those cycles is consistent."""
    ref = golden_large["hist"]
    mg = O.MGOracle(1025, 1025, dtype=np.float64, max_levels=9, cycle="V", smoother="jacobi",
                    omega=0.8, jacobi_form="vectorized")
    rhs = O.sine_rhs(1025, 1025)
    u, info = mg.solve(rhs, tol=1e-10, max_iterations=3)
    np.testing.assert_allclose(info["residual_history"], ref[:3], rtol=1e-9)
    assert len(mg.shapes) == 9 and mg.shapes[-1] == (5, 5)


@pytest.mark.parametrize("n,levels,cyc,kind,omega", [(129, 6, "V", "jacobi", 0.8), (65, 5, "W", "rbgs", 1.0), ((65, 33), 4, "V", "rbgs", 1.15),
                                                     (33, 4, "F", "jacobi", 2.0 / 3.0)])
def test_c_oracle_equals_numpy_oracle(n, levels, cyc, kind, omega):
    """oracle/mg_oracle.c (the multi-threaded CPU baseline of bench.py) reproduces the pinned NumPy oracle bit for bit."""
    from oracle.c_oracle import COracle
    nx, ny = (n, n) if isinstance(n, int) else n
    rng = np.random.default_rng(nx)
    rhs = O.sine_rhs(nx, ny) + 0.01 * rng.standard_normal((nx, ny))
    rhs[0, :] = rhs[-1, :] = rhs[:, 0] = rhs[:, -1] = 0.0          # converge the coarsest solve (stop test at round-off)
    u0 = rng.standard_normal((nx, ny))
    ref = O.MGOracle(nx, ny, max_levels=levels, cycle=cyc, smoother=kind, omega=omega, jacobi_form="vectorized")
    ref.rhs[0] = rhs.copy()
    u = u0.copy()
    co = COracle(nx, ny, max_levels=levels, cycle=cyc, smoother=kind, omega=omega)
    co.set_problem(rhs, u0)
    for _ in range(3):
        u = ref.cycle_once(u, 0)
        co.cycle()
    got = co.solution()
    assert np.max(np.abs(got - u)) <= 1e-13 * np.max(np.abs(u))      # the coarsest stop test may differ by one sweep
    np.testing.assert_allclose(co.residual_norm(), ref.residual_norm(u, rhs, 0), rtol=1e-9)
    co.close()


def test_oracle_equals_reference_at_the_bench_size(golden_large4097):
    """oracle == reference at 4097^2 (BASELINE config 3's grid): two V(2,2) Jacobi cycles, history and a strided sample."""
    n = 4097
    g = golden_large4097
    mgo = O.MGOracle(n, n, max_levels=11, cycle="V", smoother="jacobi", omega=0.8, jacobi_form="vectorized")
    rhs = O.sine_rhs(n, n)
    mgo.rhs[0] = rhs.copy()
    u = np.zeros_like(rhs)
    hist = []
    for _ in range(2):
        u = mgo.cycle_once(u, 0)
        hist.append(mgo.residual_norm(u, rhs, 0))
    np.testing.assert_allclose(hist, g["hist"], rtol=1e-12)
    assert np.max(np.abs(u[::128, ::128] - g["u_sample"])) <= 1e-13 * float(g["u_linf"])
