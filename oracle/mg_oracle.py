"""CPU oracle for the multigrid V/W-cycle hot path -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the reference's *CPU* algorithm for the
hot path.  It exists so that the HIP path can be checked
on the GPU box, where /root/reference does not exist.  Nothing in the product
package imports it: only tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py may.

Parity status: PINNED.  tests/golden/*.npz were produced by importing the
reference itself in the build container (tests/golden/generate_golden.py) in
its one self-consistent configuration -- LaplacianOperator(coefficient=-1.0),
full_weighting / bilinear transfers, MultigridSolver -- and
tests/test_oracle_golden.py checks every function below against them
(bit-for-bit for the element-wise operators, <= 1e-13 relative for histories).

All file:line citations are relative to /root/reference/src/multigrid/.

dtype semantics follow the reference as executed under NumPy >= 2 (NEP 50):
Python-float coefficients are "weak", so an fp32 array is processed in fp32.
"""

from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# a1  Grid metadata -- core/grid.py:18-58, 140-157
# --------------------------------------------------------------------------

def grid_spacing(nx, ny, domain=(0.0, 1.0, 0.0, 1.0)):
    """hx, hy as Grid.__init__ computes them (core/grid.py:43-44)."""
    if nx < 3 or ny < 3:
        raise ValueError("Grid must have at least 3 points in each direction")
    # Python floats on purpose: NumPy treats them as weak scalars, so fp32 fields stay fp32.
    return (float(domain[1]) - float(domain[0])) / (nx - 1), (float(domain[3]) - float(domain[2])) / (ny - 1)


def coarsen_shape(nx, ny):
    """Grid.coarsen (core/grid.py:140-157): needs (n-1) even."""
    if (nx - 1) % 2 != 0 or (ny - 1) % 2 != 0:
        raise ValueError("Cannot coarsen grid: need even number of interior points")
    cnx, cny = (nx - 1) // 2 + 1, (ny - 1) // 2 + 1
    if cnx < 3 or cny < 3:
        raise ValueError("Grid must have at least 3 points in each direction")
    return cnx, cny


def hierarchy_shapes(nx, ny, max_levels):
    """MultigridSolver._build_hierarchy (solvers/multigrid.py:135-171)."""
    shapes = [(nx, ny)]
    for _ in range(1, max_levels):
        try:
            c = coarsen_shape(*shapes[-1])
        except ValueError:
            break
        if c[0] < 5 or c[1] < 5:
            break
        shapes.append(c)
    return shapes


# --------------------------------------------------------------------------
# a2/a3  Operator and residual -- operators/laplacian.py:44-80, 105-124
# --------------------------------------------------------------------------

def apply_laplacian(u, hx, hy, coeff=-1.0, shift=0.0):
    """LaplacianOperator.apply: coeff * 5-point stencil, zero on the boundary.
    shift (ours, not in the reference; 0 = the reference): A = coeff * (Laplacian_h - shift I), the Helmholtz
    operator of an implicit heat step -- the shift only joins the diagonal, here and in every smoother below."""
    out = np.zeros_like(u)
    out[1:-1, 1:-1] = coeff * (
        (u[2:, 1:-1] + u[:-2, 1:-1]) / hx**2
        + (u[1:-1, 2:] + u[1:-1, :-2]) / hy**2
        - u[1:-1, 1:-1] * (2.0 / hx**2 + 2.0 / hy**2 + shift)
    )
    return out


def residual(u, f, hx, hy, coeff=-1.0, shift=0.0):
    """LaplacianOperator.residual: r = f - A u at every cell (boundary r = f)."""
    return f - apply_laplacian(u, hx, hy, coeff, shift)


# --------------------------------------------------------------------------
# a4  Norm -- core/grid.py:174-187
# --------------------------------------------------------------------------

def l2_norm(field, hx, hy):
    """Grid.l2_norm: sqrt(hx*hy*sum(field**2)) over ALL cells, in field's dtype."""
    return np.sqrt(hx * hy * np.sum(field**2))


# --------------------------------------------------------------------------
# a5  Weighted Jacobi -- solvers/smoothers.py:41-86 (loop form, omega 2/3 or
#     0.8 via WeightedJacobiSmoother :210-225) and solvers/iterative.py:72-108
#     (vectorised twin).  The two differ only in "/ hx**2" vs "* (1/hx**2)".
# --------------------------------------------------------------------------

def jacobi(u, rhs, hx, hy, omega, nu=1, form="loop", shift=0.0):
    u = u.copy()
    if form == "loop":          # smoothers.py:65-83
        diag = -2.0 / hx**2 - 2.0 / hy**2 - shift
        for _ in range(nu):
            old = u.copy()
            nb = (old[2:, 1:-1] + old[:-2, 1:-1]) / hx**2 + (old[1:-1, 2:] + old[1:-1, :-2]) / hy**2
            new = (rhs[1:-1, 1:-1] + nb) / (-diag)
            u[1:-1, 1:-1] = (1 - omega) * old[1:-1, 1:-1] + omega * new
    elif form == "vectorized":  # iterative.py:84-104
        hx2_inv, hy2_inv = 1.0 / (hx**2), 1.0 / (hy**2)
        diag = -(2.0 * hx2_inv + 2.0 * hy2_inv + shift)
        for _ in range(nu):
            old = u.copy()
            nb = hx2_inv * (old[2:, 1:-1] + old[:-2, 1:-1]) + hy2_inv * (old[1:-1, 2:] + old[1:-1, :-2])
            new = (rhs[1:-1, 1:-1] + nb) / (-diag)
            u[1:-1, 1:-1] = (1.0 - omega) * old[1:-1, 1:-1] + omega * new
    else:
        raise ValueError(form)
    return u


# --------------------------------------------------------------------------
# a6  Red-black Gauss-Seidel -- solvers/smoothers.py:117-151, 175-207
#     red = (i+j) even first, then black; same-colour points are independent,
#     so one masked vector update per colour equals the reference's loops.
# --------------------------------------------------------------------------

def _colour_mask(shape, colour):
    i = np.arange(1, shape[0] - 1)[:, None]
    j = np.arange(1, shape[1] - 1)[None, :]
    return ((i + j) % 2) == colour


def rbgs(u, rhs, hx, hy, omega=1.0, nu=1, shift=0.0):
    u = u.copy()
    diag = -2.0 / hx**2 - 2.0 / hy**2 - shift
    masks = (_colour_mask(u.shape, 0), _colour_mask(u.shape, 1))
    for _ in range(nu):
        for m in masks:
            nb = (u[2:, 1:-1] + u[:-2, 1:-1]) / hx**2 + (u[1:-1, 2:] + u[1:-1, :-2]) / hy**2
            new = (rhs[1:-1, 1:-1] + nb) / (-diag)
            upd = (1 - omega) * u[1:-1, 1:-1] + omega * new
            inner = u[1:-1, 1:-1]
            inner[m] = upd[m]
    return u


# --------------------------------------------------------------------------
# a7  Lexicographic GS (coarsest solver) -- solvers/smoothers.py:153-173 and
#     IterativeSolver.solve solvers/base.py:234-290.
#     A point (i,j) needs the NEW (i-1,j),(i,j-1) and the OLD (i+1,j),(i,j+1):
#     all points of one anti-diagonal i+j=const are mutually independent and
#     depend only on the previous (new) and next (old) anti-diagonals, so a
#     sweep over anti-diagonals reproduces the lexicographic loop exactly.
# --------------------------------------------------------------------------

def lexgs_sweep(u, rhs, hx, hy, omega=1.0, shift=0.0):
    nx, ny = u.shape
    diag = -2.0 / hx**2 - 2.0 / hy**2 - shift
    for s in range(2, nx + ny - 3):
        i = np.arange(max(1, s - (ny - 2)), min(nx - 2, s - 1) + 1)
        j = s - i
        nb = (u[i + 1, j] + u[i - 1, j]) / hx**2 + (u[i, j + 1] + u[i, j - 1]) / hy**2
        new = (rhs[i, j] + nb) / (-diag)
        u[i, j] = (1 - omega) * u[i, j] + omega * new
    return u


def coarse_solve(u0, rhs, hx, hy, coeff=-1.0, tol=1e-12, maxit=1000, omega=1.0, shift=0.0):
    """IterativeSolver.solve with the lex-GS smoother (base.py:255-290):
    sweep, residual, norm, stop when norm < tol or after maxit sweeps.
    Returns (u, sweeps)."""
    u = u0.copy()
    it = 0
    for it in range(1, maxit + 1):
        u = lexgs_sweep(u.copy(), rhs, hx, hy, omega, shift)
        if l2_norm(residual(u, rhs, hx, hy, coeff, shift), hx, hy) < tol:
            break
    return u, it


# --------------------------------------------------------------------------
# a8  Full-weighting restriction -- operators/transfer.py:100-124
#     interior 1/16 corners + 1/8 edges + 1/4 centre (in that association),
#     coarse boundary = injection; output array in out_dtype (the grid dtype).
# --------------------------------------------------------------------------

def restrict_fw(field, out_dtype=None):
    nx, ny = field.shape
    cnx, cny = (nx - 1) // 2 + 1, (ny - 1) // 2 + 1
    out = np.zeros((cnx, cny), dtype=out_dtype or field.dtype)
    out[:, :] = field[0:2 * cnx:2, 0:2 * cny:2][:cnx, :cny]          # injection everywhere
    c = field[2:-2:2, 2:-2:2]
    corners = ((field[1:-3:2, 1:-3:2] + field[1:-3:2, 3:-1:2]) + field[3:-1:2, 1:-3:2]) + field[3:-1:2, 3:-1:2]
    edges = ((field[1:-3:2, 2:-2:2] + field[3:-1:2, 2:-2:2]) + field[2:-2:2, 1:-3:2]) + field[2:-2:2, 3:-1:2]
    out[1:-1, 1:-1] = (1.0 / 16.0 * corners + 1.0 / 8.0 * edges) + 1.0 / 4.0 * c
    return out


# --------------------------------------------------------------------------
# a9  Bilinear prolongation -- operators/transfer.py:234-267
#     quirk F9: fine[odd i, ny-1] and fine[nx-1, odd j] stay 0.
#     All arithmetic happens on the fine array, i.e. in out_dtype.
# --------------------------------------------------------------------------

def prolong_bilinear(field, out_dtype=None):
    cnx, cny = field.shape
    nx, ny = 2 * (cnx - 1) + 1, 2 * (cny - 1) + 1
    fine = np.zeros((nx, ny), dtype=out_dtype or field.dtype)
    fine[0::2, 0::2] = field
    # odd i, even j, only j < ny-1  (transfer.py:250-253)
    fine[1:-1:2, 0:ny - 1:2] = 0.5 * (fine[0:-2:2, 0:ny - 1:2] + fine[2::2, 0:ny - 1:2])
    # even i, odd j, only i < nx-1  (transfer.py:256-259)
    fine[0:nx - 1:2, 1:-1:2] = 0.5 * (fine[0:nx - 1:2, 0:-2:2] + fine[0:nx - 1:2, 2::2])
    # odd i, odd j (transfer.py:262-265)
    fine[1:-1:2, 1:-1:2] = 0.25 * (
        ((fine[0:-2:2, 0:-2:2] + fine[0:-2:2, 2::2]) + fine[2::2, 0:-2:2]) + fine[2::2, 2::2]
    )
    return fine


# --------------------------------------------------------------------------
# a13  Precision policy -- core/precision.py
# --------------------------------------------------------------------------

class OraclePrecision:
    """Minimal restatement of PrecisionManager (core/precision.py:18-357):
    'double' | 'single' | 'mixed', adaptive threshold rule, per-level split."""

    def __init__(self, default="double", adaptive=True, convergence_threshold=1e-6,
                 memory_threshold_gb=4.0):
        names = {"single": "float32", "float32": "float32", "double": "float64",
                 "float64": "float64", "mixed": "mixed"}
        self.current = names[default]
        self.adaptive = adaptive
        self.thr = convergence_threshold
        self.mem = memory_threshold_gb * 1024**3
        self.history = [self.current]

    def dtype(self, p=None):                       # precision.py:85-104
        p = p or self.current
        return np.float32 if p == "float32" else np.float64

    def convert(self, a, p=None):                  # precision.py:106-134
        dt = self.dtype(p)
        return a if a.dtype == dt else a.astype(dt)

    def update(self, rnorm, shapes):               # precision.py:270-302
        if not self.adaptive:
            return False
        old = self.current
        mem = sum(a * b for a, b in shapes) * np.dtype(self.dtype()).itemsize * 4   # :136-153
        down = mem > self.mem or (self.current == "float64" and rnorm > self.thr * 100)  # :155-187
        if down:
            if self.current == "float64":
                self.current = "float32"
        elif self.current == "float32" and rnorm < self.thr * 10:                    # :248-268
            self.current = "float64"
        if self.current != old:
            self.history.append(self.current)
            return True
        return False

    def for_level(self, level, nlevels):           # precision.py:337-357
        if not self.adaptive or self.current != "mixed":
            return self.current
        return "float32" if level >= nlevels // 2 else "float64"


# --------------------------------------------------------------------------
# a10-a12  Hierarchy, cycle and outer loop -- solvers/multigrid.py:135-337
# --------------------------------------------------------------------------

class MGOracle:
    """Restatement of MultigridSolver (solvers/multigrid.py) in the oracle
    configuration of SURVEY.md section 8c."""

    def __init__(self, nx, ny, domain=(0.0, 1.0, 0.0, 1.0), dtype=np.float64, coeff=-1.0,
                 max_levels=4, cycle="V", pre=2, post=2, smoother="jacobi", omega=0.8,
                 jacobi_form="loop", coarse_tol=1e-12, coarse_maxit=1000, shift=0.0):
        self.dtype = np.dtype(dtype)
        self.coeff = coeff
        self.shift = shift                         # Helmholtz shift on every level (0: the reference's operator)
        self.cycle = cycle
        self.pre, self.post = pre, post
        self.smoother, self.omega, self.jform = smoother, omega, jacobi_form
        self.ctol, self.cmaxit = coarse_tol, coarse_maxit
        self.shapes = hierarchy_shapes(nx, ny, max_levels)
        self.h = [grid_spacing(a, b, domain) for a, b in self.shapes]
        self.rhs = [np.zeros(s, dtype=self.dtype) for s in self.shapes]
        self.coarse_sweeps = []

    # -- helpers ---------------------------------------------------------
    def _smooth(self, u, level, nu):
        hx, hy = self.h[level]
        if self.smoother == "jacobi":
            return jacobi(u, self.rhs[level], hx, hy, self.omega, nu, self.jform, self.shift)
        if self.smoother == "rbgs":
            return rbgs(u, self.rhs[level], hx, hy, self.omega, nu, self.shift)
        if self.smoother == "lexgs":
            u = u.copy()
            for _ in range(nu):
                lexgs_sweep(u, self.rhs[level], hx, hy, self.omega, self.shift)
            return u
        raise ValueError(self.smoother)

    def residual_norm(self, u, rhs, level=0):      # multigrid.py:372-375
        hx, hy = self.h[level]
        return float(l2_norm(residual(u, rhs, hx, hy, self.coeff, self.shift), hx, hy))

    # -- cycle (multigrid.py:253-337) --------------------------------------
    def cycle_once(self, u, level=0, pm=None):
        L = len(self.shapes)
        hx, hy = self.h[level]
        if level == L - 1:                         # multigrid.py:270-272, 355-370
            u, sweeps = coarse_solve(u, self.rhs[level], hx, hy, self.coeff, self.ctol, self.cmaxit, shift=self.shift)
            self.coarse_sweeps.append(sweeps)
            return u
        if pm is not None:                         # multigrid.py:275-285
            p = pm.for_level(level, L)
            u = pm.convert(u, p)
            self.rhs[level] = pm.convert(self.rhs[level], p)
        if self.pre > 0:
            u = self._smooth(u, level, self.pre)
        r = residual(u, self.rhs[level], hx, hy, self.coeff, self.shift)
        self.rhs[level + 1] = restrict_fw(r, self.dtype).copy()     # :298-304
        e = np.zeros_like(self.rhs[level + 1])
        if self.cycle == "V":
            reps = 1
        elif self.cycle == "W":
            reps = 2
        else:                                      # 'F', multigrid.py:315-319
            reps = max(1, 2 ** (L - level - 2))
        for _ in range(reps):
            e = self.cycle_once(e, level + 1, pm)
        fine = prolong_bilinear(e, self.dtype)     # :323-325 (fine grid dtype)
        u += fine                                  # :329
        if self.post > 0:
            u = self._smooth(u, level, self.post)
        return u

    # -- full-multigrid initial guess (advanced_multigrid.py:626-683) ---------
    def fmg_init(self, rhs, cycles=1, ring=None):
        """Restrict rhs to every level (full weighting), solve the coarsest level from zero, then per level upward:
        u = P u_coarse followed by `cycles` cycles of the sub-hierarchy starting at that level.  `ring`: Dirichlet
        data kept on the finest boundary (our extension; the reference overwrites it with zeros)."""
        L = len(self.shapes)
        hier = [rhs.copy()]
        for _ in range(L - 1):
            hier.append(restrict_fw(hier[-1], self.dtype))
        self.rhs[L - 1] = hier[-1].copy()
        hx, hy = self.h[L - 1]
        u, _ = coarse_solve(np.zeros_like(hier[-1]), hier[-1], hx, hy, self.coeff, self.ctol, self.cmaxit, shift=self.shift)
        for level in range(L - 2, -1, -1):
            u = prolong_bilinear(u, self.dtype)
            if level == 0 and ring is not None:
                u[0, :], u[-1, :], u[:, 0], u[:, -1] = ring[0, :], ring[-1, :], ring[:, 0], ring[:, -1]
            self.rhs[level] = hier[level].copy()
            for _ in range(cycles):
                u = self.cycle_once(u, level)
        return u

    # -- outer loop (multigrid.py:184-251) ---------------------------------
    def solve(self, rhs, u0=None, tol=1e-8, max_iterations=50, pm=None):
        u = np.zeros_like(rhs) if u0 is None else u0.copy()
        self.rhs[0] = rhs.copy()
        hist, prec = [], []
        converged = False
        it = 0
        for it in range(1, max_iterations + 1):
            if pm is not None:
                pm.update(self.residual_norm(u, rhs, 0), self.shapes)
            u = self.cycle_once(u, 0, pm)
            rn = self.residual_norm(u, rhs, 0)
            hist.append(rn)
            prec.append(pm.current if pm is not None else "double")
            if rn < tol:
                converged = True
                break
        return u, {"converged": converged, "iterations": it, "final_residual": hist[-1],
                   "residual_history": hist, "precision_levels": prec,
                   "num_levels": len(self.shapes), "grid_hierarchy": list(self.shapes)}


# --------------------------------------------------------------------------
# Mixed-precision residual and defect correction (gpu/cuda_kernels.py:843-883, 915-929, 937-967 are the reference's
# building blocks: fp32 iterate / rhs in, fp64 residual out; a correction applied across precisions).  The reference
# never assembles them into a working solver (its kernels carry the sign errors of SURVEY F5), so the LOOP below is
# our design ("parity unpinned"); every operator inside it is one of the pinned restatements above.
# --------------------------------------------------------------------------

def residual_mixed(u32, f32, hx, hy, coeff=-1.0):
    """fp64 residual of an fp32 iterate and rhs: operands up-cast, then `residual` (boundary r = f)."""
    return residual(np.asarray(u32, dtype=np.float32).astype(np.float64), np.asarray(f32, dtype=np.float32).astype(np.float64),
                    hx, hy, coeff)


def defect_correction(mgo, rhs, u0=None, tol=1e-8, max_iterations=50):
    """Iterative refinement around an MGOracle built on a float64 grid: fp64 iterate and residual, one cycle of `mgo`
    under PrecisionManager('single', adaptive=False) (every level but the coarsest in fp32) from the zero correction on
    A e = r per outer step, u += e in fp64.  The fp32 right-hand side carries a zero boundary ring (the Dirichlet
    data of u are exact); the recorded norm is the reference's (boundary cells r = f)."""
    pm = OraclePrecision("single", adaptive=False)
    hx, hy = mgo.h[0]
    u = np.zeros_like(rhs) if u0 is None else u0.astype(np.float64).copy()
    hist = []
    r = residual(u, rhs, hx, hy, mgo.coeff, mgo.shift)
    r0 = float(l2_norm(r, hx, hy))
    for _ in range(max_iterations):
        r32 = r.astype(np.float32)
        r32[0, :] = r32[-1, :] = 0.0
        r32[:, 0] = r32[:, -1] = 0.0
        mgo.rhs[0] = r32
        e = mgo.cycle_once(np.zeros_like(rhs), 0, pm)
        u = u + e.astype(np.float64)
        r = residual(u, rhs, hx, hy, mgo.coeff, mgo.shift)
        hist.append(float(l2_norm(r, hx, hy)))
        if hist[-1] < tol:
            break
    return u, {"initial_residual": r0, "residual_history": hist, "iterations": len(hist), "converged": hist[-1] < tol}


# --------------------------------------------------------------------------
# Variable-coefficient operator  A u = coeff * div(a grad u)  -- NOT in the reference (README
# bullet only).  PARITY UNPINNED: this is a restatement of OUR discretisation (csrc/mg_kernels.hpp varcoef_kernel),
# checked by (i) a == 1 reproducing the constant-coefficient functions above bit for bit on dyadic grids and
# (ii) second-order convergence on a manufactured solution.
#   face values: arithmetic means of the vertex values; coarse operators: a injected (re-discretisation).
#   Round 3: the smoothers MULTIPLY by the reciprocal diagonal, un = (f + nb) * (1 / D) with 1 / D rounded once in the level's
#   dtype, instead of dividing by D -- the diagonal depends on the coefficient only, so the device keeps 1 / D as one more field
#   per level and its sweeps have no division left (they were division-bound).  For a == 1 on dyadic grids 1 / D is exact and
#   nothing changes; elsewhere results move by an ulp (this operator is ours to define: there is no reference to keep).
# --------------------------------------------------------------------------

def _faces(a):
    c = a[1:-1, 1:-1]
    return 0.5 * (c + a[2:, 1:-1]), 0.5 * (c + a[:-2, 1:-1]), 0.5 * (c + a[1:-1, 2:]), 0.5 * (c + a[1:-1, :-2])


def var_residual(u, f, a, hx, hy, coeff=-1.0, shift=0.0):
    ihx2, ihy2 = 1.0 / (hx * hx), 1.0 / (hy * hy)
    aip, aim, ajp, ajm = _faces(a)
    sx = aip * u[2:, 1:-1] + aim * u[:-2, 1:-1]
    sy = ajp * u[1:-1, 2:] + ajm * u[1:-1, :-2]
    D = (aip + aim) * ihx2 + (ajp + ajm) * ihy2 + shift
    r = f.copy()
    r[1:-1, 1:-1] = f[1:-1, 1:-1] - coeff * ((sx * ihx2 + sy * ihy2) - u[1:-1, 1:-1] * D)
    return r


def _var_update(u, f, a, hx, hy, omega, shift=0.0):
    ihx2, ihy2 = 1.0 / (hx * hx), 1.0 / (hy * hy)
    aip, aim, ajp, ajm = _faces(a)
    sx = aip * u[2:, 1:-1] + aim * u[:-2, 1:-1]
    sy = ajp * u[1:-1, 2:] + ajm * u[1:-1, :-2]
    D = (aip + aim) * ihx2 + (ajp + ajm) * ihy2 + shift
    un = (f[1:-1, 1:-1] + (ihx2 * sx + ihy2 * sy)) * (1.0 / D)
    return (1.0 - omega) * u[1:-1, 1:-1] + omega * un


def var_jacobi(u, f, a, hx, hy, omega, nu=1, shift=0.0):
    u = u.copy()
    for _ in range(nu):
        u[1:-1, 1:-1] = _var_update(u, f, a, hx, hy, omega, shift)
    return u


def var_rbgs(u, f, a, hx, hy, omega=1.0, nu=1, shift=0.0):
    u = u.copy()
    masks = (_colour_mask(u.shape, 0), _colour_mask(u.shape, 1))
    for _ in range(nu):
        for m in masks:
            upd = _var_update(u, f, a, hx, hy, omega, shift)
            inner = u[1:-1, 1:-1]
            inner[m] = upd[m]
    return u


def var_lexgs_sweep(u, f, a, hx, hy, shift=0.0):
    nx, ny = u.shape
    hx2, hy2 = hx * hx, hy * hy
    for s in range(2, nx + ny - 3):
        i = np.arange(max(1, s - (ny - 2)), min(nx - 2, s - 1) + 1)
        j = s - i
        aip, aim = 0.5 * (a[i, j] + a[i + 1, j]), 0.5 * (a[i, j] + a[i - 1, j])
        ajp, ajm = 0.5 * (a[i, j] + a[i, j + 1]), 0.5 * (a[i, j] + a[i, j - 1])
        nb = (aip * u[i + 1, j] + aim * u[i - 1, j]) / hx2 + (ajp * u[i, j + 1] + ajm * u[i, j - 1]) / hy2
        D = (aip + aim) / hx2 + (ajp + ajm) / hy2 + shift
        u[i, j] = (1 - 1.0) * u[i, j] + 1.0 * ((f[i, j] + nb) * (1.0 / D))
    return u


def var_coarse_residual(u, f, a, hx, hy, coeff, shift=0.0):
    hx2, hy2 = hx * hx, hy * hy
    aip, aim, ajp, ajm = _faces(a)
    D = (aip + aim) / hx2 + (ajp + ajm) / hy2 + shift
    r = f.copy()
    r[1:-1, 1:-1] = f[1:-1, 1:-1] - coeff * (((aip * u[2:, 1:-1] + aim * u[:-2, 1:-1]) / hx2 +
                                               (ajp * u[1:-1, 2:] + ajm * u[1:-1, :-2]) / hy2) - u[1:-1, 1:-1] * D)
    return r


class VarMGOracle(MGOracle):
    """MGOracle with A = coeff * div(a grad .): same cycle, variable-coefficient smoother / residual / coarsest solve.
    The coefficient of level l is every 2^l-th vertex value of the caller's array (re-discretisation), cast once to the
    precision the level computes in (csrc/mghip.hip mg_set_coefficient does the same)."""

    def __init__(self, a, *args, **kw):
        super().__init__(a.shape[0], a.shape[1], *args, **kw)
        self.a_src = np.asarray(a)
        self._a_cache = {}
        self.a = [self.a_of(l, self.dtype) for l in range(len(self.shapes))]

    def a_of(self, level, dtype):
        key = (level, np.dtype(dtype).str)
        if key not in self._a_cache:
            self._a_cache[key] = np.ascontiguousarray(self.a_src[::2**level, ::2**level]).astype(dtype)
        return self._a_cache[key]

    def _smooth(self, u, level, nu):
        hx, hy = self.h[level]
        a = self.a_of(level, u.dtype)
        if self.smoother == "jacobi":
            return var_jacobi(u, self.rhs[level], a, hx, hy, self.omega, nu, self.shift)
        return var_rbgs(u, self.rhs[level], a, hx, hy, self.omega, nu, self.shift)

    def residual_norm(self, u, rhs, level=0):
        hx, hy = self.h[level]
        return float(l2_norm(var_residual(u, rhs, self.a_of(level, u.dtype), hx, hy, self.coeff, self.shift), hx, hy))

    def cycle_once(self, u, level=0, pm=None):
        L = len(self.shapes)
        hx, hy = self.h[level]
        if level == L - 1:                         # never converted: the grid dtype (solvers/multigrid.py:270-272)
            u = u.copy()
            a = self.a_of(level, u.dtype)
            sweeps = self.cmaxit
            for it in range(1, self.cmaxit + 1):
                var_lexgs_sweep(u, self.rhs[level], a, hx, hy, self.shift)
                if l2_norm(var_coarse_residual(u, self.rhs[level], a, hx, hy, self.coeff, self.shift), hx, hy) < self.ctol:
                    sweeps = it
                    break
            self.coarse_sweeps.append(sweeps)
            return u
        if pm is not None:                         # multigrid.py:275-285
            p = pm.for_level(level, L)
            u = pm.convert(u, p)
            self.rhs[level] = pm.convert(self.rhs[level], p)
        u = self._smooth(u, level, self.pre)
        r = var_residual(u, self.rhs[level], self.a_of(level, u.dtype), hx, hy, self.coeff, self.shift)
        self.rhs[level + 1] = restrict_fw(r, self.dtype).copy()
        e = np.zeros_like(self.rhs[level + 1])
        reps = 1 if self.cycle == "V" else 2 if self.cycle == "W" else max(1, 2 ** (L - level - 2))
        for _ in range(reps):
            e = self.cycle_once(e, level + 1, pm)
        u += prolong_bilinear(e, self.dtype)
        return self._smooth(u, level, self.post)


def sine_rhs(nx, ny, domain=(0.0, 1.0, 0.0, 1.0), dtype=np.float64):
    """f = 2 pi^2 sin(pi x) sin(pi y) on Grid.X/Grid.Y (README.md:77-78,
    gpu/gpu_benchmark.py:179-184); linspace in `dtype` as core/grid.py:48-50."""
    x = np.linspace(domain[0], domain[1], nx, dtype=dtype)
    y = np.linspace(domain[2], domain[3], ny, dtype=dtype)
    X, Y = np.meshgrid(x, y, indexing="ij")
    return 2 * np.pi**2 * np.sin(np.pi * X) * np.sin(np.pi * Y)
