"""Row a14 of SURVEY 8(a): the reference's MixedPrecisionKernels interface (fp32 iterate -> fp64 residual) and the
defect-correction policy built on it (MG_PREC_DEFECT: fp64 iterate and residual, fp32 cycles on the error equation).
The mixed residual is pinned through the pinned fp64 residual; the defect LOOP is our own design (the reference never
assembles one): parity unpinned, checked against oracle.defect_correction and against the fp64 solve."""
import numpy as np
import pytest

import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
from oracle import mg_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(17, 17), (65, 129), (513, 257), (1025, 1025)])
def test_mixed_precision_residual_kernel(shape):
    """MixedPrecisionKernels.compute_mixed_precision_residual (gpu/cuda_kernels.py:937-967): bit-exact against the fp64
    residual of the up-cast operands; boundary cells r = f."""
    nx, ny = shape
    rng = np.random.default_rng(nx * 7 + ny)
    u = rng.standard_normal(shape).astype(np.float32)
    f = rng.standard_normal(shape).astype(np.float32)
    hx, hy = O.grid_spacing(nx, ny)
    r = mg.MixedPrecisionKernels().compute_mixed_precision_residual(u, f, hx, hy)
    assert r.dtype == np.float64
    np.testing.assert_array_equal(r, O.residual_mixed(u, f, hx, hy))
    np.testing.assert_array_equal(r[0, :], f[0, :].astype(np.float64))
    # it is NOT the fp32 residual promoted afterwards
    r32 = mg.LaplacianOperator(-1.0).residual(mg.Grid(nx, ny, dtype=np.float32), u, f)
    assert np.max(np.abs(r - r32)) > 0


def test_kernel_wrapper_classes_compute_the_cpu_plugins():
    """SmoothingKernels / TransferKernels (gpu/cuda_kernels.py:284-436, 738-828): reference names, the CPU path's numbers."""
    n = 33
    rng = np.random.default_rng(1)
    u = rng.standard_normal((n, n)); f = rng.standard_normal((n, n))
    hx, hy = O.grid_spacing(n, n)
    sk, tk = mg.SmoothingKernels(), mg.TransferKernels()
    out = np.zeros_like(u)
    sk.jacobi_smoothing(u, out, f, hx, hy, num_iterations=3, relaxation_parameter=0.8)
    np.testing.assert_array_equal(out, O.jacobi(u, f, hx, hy, 0.8, 3, "vectorized"))
    v = u.copy(); sk.red_black_gauss_seidel(v, f, hx, hy, 2)
    np.testing.assert_array_equal(v, O.rbgs(u, f, hx, hy, 1.0, 2))
    r = np.empty_like(u); tk.compute_residual(u, f, r, hx, hy)
    np.testing.assert_array_equal(r, O.residual(u, f, hx, hy, -1.0))
    c = np.empty((17, 17)); tk.restriction(r, c)
    np.testing.assert_array_equal(c, O.restrict_fw(r, np.float64))
    fine = np.full((n, n), 7.0); tk.prolongation(c, fine)
    np.testing.assert_array_equal(fine, O.prolong_bilinear(c, np.float64))
    with pytest.raises(ValueError):
        tk.restriction(r, np.empty((16, 17)))


@pytest.mark.parametrize("fused", [2, 1, 0])
@pytest.mark.parametrize("n,cyc,kind,omega", [(129, "V", "jacobi", 0.8), (257, "V", "rbgs", 1.0), (65, "W", "rbgs", 1.15)])
def test_defect_correction_equals_oracle(n, cyc, kind, omega, fused):
    levels = mg.default_max_levels(n, n)
    rng = np.random.default_rng(n)
    rhs = O.sine_rhs(n, n) + 0.01 * rng.standard_normal((n, n))
    u0 = np.zeros((n, n)); u0[0, :] = rng.standard_normal(n); u0[:, -1] = rng.standard_normal(n)      # Dirichlet data
    mgo = O.MGOracle(n, n, max_levels=levels, cycle=cyc, smoother=kind, omega=omega, jacobi_form="vectorized")
    u_ref, info = O.defect_correction(mgo, rhs, u0, tol=0.0, max_iterations=6)
    eng = mg.MultigridEngine(n, n, max_levels=levels, cycle=cyc, smoother=_lib.MG_JACOBI if kind == "jacobi" else _lib.MG_RBGS,
                             omega=omega, precision=_lib.MG_PREC_DEFECT, fused=fused)
    u, r = eng.solve(rhs, u0, tol=0.0, max_iterations=6)
    # device-resident stepping gives the same iterate
    eng.set_rhs(rhs); eng.set_solution(u0); eng.cycle(6)
    u_step = eng.get_solution()
    n_step = eng.residual_norm()
    eng.close()
    assert r["precision_codes"] == [3] * 6
    np.testing.assert_allclose(r["initial_residual"], info["initial_residual"], rtol=1e-12)
    np.testing.assert_allclose(r["residual_history"], info["residual_history"], rtol=1e-7, atol=1e-13)
    assert np.max(np.abs(u - u_ref)) <= 1e-12 * np.max(np.abs(u_ref))
    np.testing.assert_array_equal(u_step, u)
    np.testing.assert_allclose(n_step, r["residual_history"][-1], rtol=1e-12)


@pytest.mark.parametrize("n", [1025, 4097])
def test_defect_correction_reaches_the_fp64_floor(n):
    """fp32 cycles + fp64 defect: the iterate converges to the fp64 solution (north_star: mixed within 1e-5; here far
    closer) and the residual to the fp64 run's floor -- where plain fp32 stalls orders of magnitude above it."""
    f = lambda x, y: 2 * np.pi**2 * np.sin(np.pi * x) * np.sin(np.pi * y)
    prob = mg.PoissonProblem(f, nx=n, ny=n)
    u64, i64 = mg.MixedPrecisionMultigrid("double", tolerance=0.0, max_iterations=30).solve(prob)
    ud, idf = mg.MixedPrecisionMultigrid("defect", tolerance=0.0, max_iterations=30).solve(prob)
    floor64 = np.median(i64["residual_history"][-5:])
    floord = np.median(idf["residual_history"][-5:])
    assert floord < 3.0 * floor64, (floord, floor64)
    assert np.max(np.abs(ud - u64)) / np.max(np.abs(u64)) < 1e-9
    assert set(idf["precision_levels_used"]) == {"defect"}
