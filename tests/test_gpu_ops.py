"""GPU parity of every operator on the path, through the C ABI (host-pointer entry points),
against (a) the golden vectors the reference produced and (b) the oracle on fresh seeded inputs.

Bar: bit-exact on grids whose h^2 is a power of two (every 2^k+1 grid on a unit domain) -- the
kernels keep the reference's rounding sequence; <= 4 ulp on the non-dyadic case, where the kernels
multiply by reciprocals instead of dividing."""
import numpy as np
import pytest

import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from oracle import mg_oracle as O

pytestmark = pytest.mark.gpu

OPS_CASES = ["sq17", "sq33", "rect17x33", "rect33x9", "nondyadic21x13", "min5"]
NEG = mg.LaplacianOperator(coefficient=-1.0)


def _grid(g, tag, dt):
    k = f"{tag}_{dt}"
    dom = tuple(float(v) for v in g[f"{k}__domain"])
    u = g[f"{k}__u"]
    return k, mg.Grid(u.shape[0], u.shape[1], dom, u.dtype), u, g[f"{k}__f"]


def _cmp(actual, desired, exact, dt):
    assert actual.dtype == desired.dtype
    if exact:
        np.testing.assert_array_equal(actual, desired)
    else:
        eps = np.finfo(desired.dtype).eps
        scale = np.max(np.abs(desired))
        assert np.max(np.abs(actual - desired)) <= 8 * eps * scale


@pytest.mark.parametrize("dt", ["float64", "float32"])
@pytest.mark.parametrize("tag", OPS_CASES)
def test_ops_match_reference_golden(golden_ops, tag, dt):
    g = golden_ops
    k, grid, u, f = _grid(g, tag, dt)
    exact = "nondyadic" not in tag
    _cmp(NEG.apply(grid, u), g[f"{k}__apply"], exact, dt)
    _cmp(NEG.residual(grid, u, f), g[f"{k}__residual"], exact, dt)
    _cmp(mg.JacobiSmoother().smooth(grid, NEG, u, f, 1), g[f"{k}__jacobi_w23_nu1"], exact, dt)
    _cmp(mg.WeightedJacobiSmoother().smooth(grid, NEG, u, f, 2), g[f"{k}__jacobi_w08_nu2"], exact, dt)
    _cmp(mg.EnhancedJacobiSolver(relaxation_parameter=0.8).smooth(grid, NEG, u, f, 2), g[f"{k}__vjacobi_w08_nu2"], exact, dt)
    _cmp(mg.GaussSeidelSmoother(red_black=True).smooth(grid, NEG, u, f, 1), g[f"{k}__rbgs_w10_nu1"], exact, dt)
    _cmp(mg.GaussSeidelSmoother(red_black=True, relaxation_parameter=1.5).smooth(grid, NEG, u, f, 2),
         g[f"{k}__rbgs_w15_nu2"], exact, dt)
    _cmp(mg.GaussSeidelSmoother().smooth(grid, NEG, u, f, 1), g[f"{k}__lexgs_w10_nu1"], exact, dt)
    tol = 1e-6 if dt == "float32" else 1e-13      # fp64 accumulation on the device vs NumPy's pairwise sum
    np.testing.assert_allclose(grid.l2_norm(g[f"{k}__residual"]), float(g[f"{k}__norm"]), rtol=tol)
    if f"{k}__restrict_fw" in g.files:
        gc = grid.coarsen()
        np.testing.assert_array_equal(mg.RestrictionOperator("full_weighting").apply(grid, u, gc), g[f"{k}__restrict_fw"])
        np.testing.assert_array_equal(mg.ProlongationOperator("bilinear").apply(gc, g[f"{k}__e"], grid), g[f"{k}__prolong"])


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(65, 65), (129, 257), (257, 129), (513, 513), (1025, 1025), (33, 2049)])
def test_ops_match_oracle_on_tile_boundaries(shape, dt):
    """Sizes that span several LDS tiles in both directions (tile = 32 rows x 512 bytes), including
    ragged last tiles; bit-exact against the oracle."""
    nx, ny = shape
    rng = np.random.default_rng(nx * 7919 + ny)
    u = rng.standard_normal(shape).astype(dt)
    f = rng.standard_normal(shape).astype(dt)
    grid = mg.Grid(nx, ny, dtype=dt)
    hx, hy = O.grid_spacing(nx, ny)
    np.testing.assert_array_equal(NEG.residual(grid, u, f), O.residual(u, f, hx, hy, -1.0))
    np.testing.assert_array_equal(mg.JacobiSmoother(relaxation_parameter=0.8).smooth(grid, NEG, u, f, 3),
                                  O.jacobi(u, f, hx, hy, 0.8, 3, "vectorized"))
    np.testing.assert_array_equal(mg.GaussSeidelSmoother(red_black=True, relaxation_parameter=1.15).smooth(grid, NEG, u, f, 2),
                                  O.rbgs(u, f, hx, hy, 1.15, 2))
    gc = grid.coarsen()
    np.testing.assert_array_equal(mg.RestrictionOperator().apply(grid, u, gc), O.restrict_fw(u))
    e = rng.standard_normal(gc.shape).astype(dt)
    np.testing.assert_array_equal(mg.ProlongationOperator().apply(gc, e, grid), O.prolong_bilinear(e))
    np.testing.assert_allclose(grid.l2_norm(u), float(O.l2_norm(u.astype(np.float64), hx, hy)), rtol=1e-13)


def test_mixed_dtype_transfers():
    """Transfers across the fp64/fp32 level boundary of the per-level MIXED policy
    (solvers/multigrid.py:281-285 + operators/transfer.py:71,207)."""
    rng = np.random.default_rng(5)
    r64 = rng.standard_normal((65, 65))
    r32 = r64.astype(np.float32)
    g64, g32 = mg.Grid(65, 65), mg.Grid(65, 65, dtype=np.float32)
    # fp32 residual restricted into an fp64 grid: arithmetic in fp32, stored as fp64
    out = mg.RestrictionOperator().apply(g64, r32, g64.coarsen())
    np.testing.assert_array_equal(out, O.restrict_fw(r32, np.float64))
    # fp32 coarse correction prolongated onto an fp64 grid: arithmetic in fp64
    e32 = rng.standard_normal((33, 33)).astype(np.float32)
    out = mg.ProlongationOperator().apply(g64.coarsen(), e32, g64)
    np.testing.assert_array_equal(out, O.prolong_bilinear(e32, np.float64))
    assert g32.coarsen().dtype == np.float32


def test_coarse_solver_matches_oracle():
    """Coarsest-grid lexicographic GS to 1e-12 (solvers/multigrid.py:119-124): same sweep count,
    same bits as the oracle's anti-diagonal restatement."""
    import ctypes as C
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
    rng = np.random.default_rng(11)
    for n in (5, 9, 17):
        rhs = rng.standard_normal((n, n))
        u0 = np.zeros((n, n))
        hx, hy = O.grid_spacing(n, n)
        want, sweeps = O.coarse_solve(u0, rhs, hx, hy, -1.0, 1e-12, 1000)
        out = np.empty_like(u0)
        got_sweeps = C.c_int(0)
        _lib.check(_lib.load().mg_op_coarse_solve(_lib.MG_F64, n, n, hx, hy, -1.0, 1e-12, 1000, _lib.ptr(u0),
                                                  _lib.ptr(rhs), _lib.ptr(out), C.byref(got_sweeps)))
        # with a non-zero boundary rhs the norm never drops below ||f_boundary||: both run 1000 sweeps
        assert got_sweeps.value == sweeps
        np.testing.assert_array_equal(out, want)
        rhs[0, :] = rhs[-1, :] = rhs[:, 0] = rhs[:, -1] = 0.0
        want, sweeps = O.coarse_solve(u0, rhs, hx, hy, -1.0, 1e-12, 1000)
        _lib.check(_lib.load().mg_op_coarse_solve(_lib.MG_F64, n, n, hx, hy, -1.0, 1e-12, 1000, _lib.ptr(u0),
                                                  _lib.ptr(rhs), _lib.ptr(out), C.byref(got_sweeps)))
        assert abs(got_sweeps.value - sweeps) <= 1          # the stop test sits at round-off of the norm
        assert np.max(np.abs(out - want)) <= 1e-13 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ring", ["zero", "nonzero"])
@pytest.mark.parametrize("domain", [(0.0, 1.0, 0.0, 1.0), (0.0, 0.75, -0.2, 1.3)])
def test_5x5_coarsest_solve_exact_sweep_counts(dtype, ring, domain):
    """The 5 x 5 end of every 2^k + 1 hierarchy, after exactly m = 1..9 sweeps (tol = 0 never stops early): the in-register
    pipelined solvers -- zero ring: 4-wide lane grid, constant update masks (lexgs_5x5_zero_ring); non-zero ring: per-lane
    time counters and ring selects (lexgs_pipelined_5x5) -- against the oracle's sweep, bit for bit, on dyadic and
    non-dyadic spacings (exact reciprocals / true divisions)."""
    import ctypes as C
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
    rng = np.random.default_rng(5)
    hx, hy = (domain[1] - domain[0]) / 4, (domain[3] - domain[2]) / 4
    code = _lib.MG_F32 if dtype == np.float32 else _lib.MG_F64
    for trial in range(4):
        rhs = rng.standard_normal((5, 5)).astype(dtype)
        u0 = rng.standard_normal((5, 5)).astype(dtype)
        if ring == "zero":
            u0[0, :] = u0[-1, :] = u0[:, 0] = u0[:, -1] = 0
        if trial == 3:
            u0[1:-1, 1:-1] = 0
        for m in range(1, 10):
            want, sweeps = O.coarse_solve(u0, rhs, hx, hy, -1.0, 0.0, m)
            out = np.empty_like(u0)
            got = C.c_int(0)
            _lib.check(_lib.load().mg_op_coarse_solve(code, 5, 5, hx, hy, -1.0, 0.0, m, _lib.ptr(u0), _lib.ptr(rhs), _lib.ptr(out),
                                                      C.byref(got)))
            assert got.value == sweeps == m
            np.testing.assert_array_equal(out, want)


def test_error_mapping_matches_reference():
    g = mg.Grid(17, 17)
    with pytest.raises(ValueError):
        NEG.apply(g, np.zeros((16, 17)))                        # operators/laplacian.py:61-62
    with pytest.raises(ValueError):
        mg.RestrictionOperator().apply(g, np.zeros((17, 17)), mg.Grid(8, 8))   # transfer.py:65-66
    with pytest.raises(ValueError):
        mg.Grid(18, 18).coarsen()                               # core/grid.py:148-149
    with pytest.raises(ValueError):
        mg.MultigridSolver().solve(g, NEG, np.zeros((17, 17)))  # solve before setup: multigrid.py:205-206


def test_size_independent_properties_at_bench_size():
    """At BASELINE's 4097^2 (too big for the oracle in a test): properties of the operators."""
    n = 4097
    grid = mg.Grid(n, n, dtype=np.float32)
    x = np.linspace(0, 1, n, dtype=np.float64)
    X = x[:, None] * np.ones((1, n))
    lin = (2.0 * X + 3.0 * X.T + 1.0).astype(np.float32)
    # Laplacian of a linear field vanishes up to fp32 rounding amplified by 1/h^2 (tests/unit/test_operators.py:57-70)
    au = NEG.apply(grid, lin)
    assert np.all(au[0, :] == 0) and np.all(au[:, -1] == 0)
    assert np.max(np.abs(au[1:-1, 1:-1])) < 6.0 * 2 ** -23 * 4 * (n - 1) ** 2
    # residual is affine in f: r(u, f1 + f2) - r(u, f1) == f2 exactly where it is representable
    ones = np.ones((n, n), dtype=np.float32)
    zero = np.zeros((n, n), dtype=np.float32)
    np.testing.assert_array_equal(NEG.residual(grid, zero, ones), ones)
    # FW restriction preserves constants (test_operators.py:205-218), prolongation of a constant is
    # constant except the reference's far-edge zeros (F9)
    gc = grid.coarsen()
    np.testing.assert_array_equal(mg.RestrictionOperator().apply(grid, ones, gc), np.ones(gc.shape, np.float32))
    p = mg.ProlongationOperator().apply(gc, np.ones(gc.shape, np.float32), grid)
    assert np.all(p[:-1, :-1] == 1) and np.all(p[1::2, -1] == 0) and np.all(p[-1, 1::2] == 0) and np.all(p[::2, -1] == 1)
    # Jacobi leaves an exact discrete solution (u = 0, f = 0) and the boundary ring untouched
    u = np.zeros((n, n), dtype=np.float32); u[0, :] = 1; u[-1, :] = 2; u[:, 0] = 3; u[:, -1] = 4
    s = mg.JacobiSmoother(relaxation_parameter=0.8).smooth(grid, NEG, u, zero, 2)
    np.testing.assert_array_equal(s[0, :], u[0, :]); np.testing.assert_array_equal(s[:, -1], u[:, -1])
    np.testing.assert_array_equal(s[-1, :], u[-1, :]); np.testing.assert_array_equal(s[:, 0], u[:, 0])
    assert np.all(s[3:-3, 3:-3] == 0)
    # norm of ones: sqrt(hx*hy*n^2)
    np.testing.assert_allclose(grid.l2_norm(ones), n / (n - 1), rtol=1e-12)


@pytest.mark.parametrize("shape", [(4097, 4097), (4500, 4097)])
def test_single_sweep_on_hbm_sized_arrays_equals_oracle(shape):
    """Arrays beyond the Infinity Cache (4097^2 fp64 = 136 MB each) take the register-blocked single sweep with streaming
    hints (jacobi_rb) instead of the LDS-tiled jacobi_kernel: same bits as the oracle, ring passed through."""
    nx, ny = shape
    rng = np.random.default_rng(nx)
    u = rng.standard_normal(shape); f = rng.standard_normal(shape)
    hx, hy = O.grid_spacing(nx, ny)
    grid = mg.Grid(nx, ny)
    got = mg.JacobiSmoother(relaxation_parameter=0.8).smooth(grid, NEG, u, f, 3)
    np.testing.assert_array_equal(got, O.jacobi(u, f, hx, hy, 0.8, 3, "vectorized"))
