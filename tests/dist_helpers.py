"""Test infrastructure for the domain-decomposition driver: a NumPy stand-in for libmghip's device-pointer
kernels with the SAME sub-domain semantics (ring pass-through, physical-side flags, windows), built from the
oracle's arithmetic.  Lets the decomposition / halo / agglomeration logic run on CPU tensors under gloo."""
import numpy as np
import torch

from oracle import mg_oracle as O

SIDE_ILO, SIDE_IHI, SIDE_JLO, SIDE_JHI = 1, 2, 4, 8


class NumpyOps:
    def __init__(self, dtype=np.float64, mixed=False):
        """mixed: per-level mixed precision on a float64 Grid (the driver allocates coarse levels in fp32): interpolation
        in fp64, like HipOps(mixed=True)."""
        self.torch = torch
        self.np_dtype = np.dtype(dtype)
        self.tdtype = torch.float32 if self.np_dtype == np.float32 else torch.float64
        self.mixed = bool(mixed)
        self.comp_dtype = np.dtype(np.float64) if self.mixed else self.np_dtype
        self._mg = None
        self._coef = None

    def alloc(self, lnx, lny, dtype=None):
        td = self.tdtype if dtype is None else (torch.float32 if np.dtype(dtype) == np.float32 else torch.float64)
        return torch.zeros((lnx, lny + 3), dtype=td)      # a few pad columns, like the pitched device layout

    @staticmethod
    def _v(t, lnx, lny):
        return t.numpy()[:lnx, :lny]

    def jacobi(self, u, rhs, out, lnx, lny, hx, hy, omega):
        res = O.jacobi(self._v(u, lnx, lny), self._v(rhs, lnx, lny), hx, hy, omega, 1, "vectorized")
        self._v(out, lnx, lny)[1:-1, 1:-1] = res[1:-1, 1:-1]       # the kernel writes owned cells only

    def rbgs_colour(self, u, rhs, lnx, lny, hx, hy, omega, colour, offset):
        a, f = self._v(u, lnx, lny), self._v(rhs, lnx, lny)
        diag = -2.0 / hx**2 - 2.0 / hy**2
        i = np.arange(1, lnx - 1)[:, None]; j = np.arange(1, lny - 1)[None, :]
        m = ((i + j + offset) % 2) == colour
        nb = (a[2:, 1:-1] + a[:-2, 1:-1]) / hx**2 + (a[1:-1, 2:] + a[1:-1, :-2]) / hy**2
        upd = (1 - omega) * a[1:-1, 1:-1] + omega * ((f[1:-1, 1:-1] + nb) / (-diag))
        inner = a[1:-1, 1:-1]
        inner[m] = upd[m]

    def residual(self, u, f, r, lnx, lny, hx, hy, coeff):
        self._v(r, lnx, lny)[...] = O.residual(self._v(u, lnx, lny), self._v(f, lnx, lny), hx, hy, coeff)

    def sumsq(self, field, i_lo, i_hi, j_lo, j_hi):
        w = field.numpy()[i_lo:i_hi, j_lo:j_hi].astype(np.float64)
        return torch.tensor([float(np.sum(w * w))], dtype=torch.float64)

    def restrict(self, fine, coarse, lnxf, lnyf, lnxc, lnyc, sides):
        f, c = self._v(fine, lnxf, lnyf), self._v(coarse, lnxc, lnyc)
        ic = np.arange(1, lnxc - 1); jc = np.arange(1, lnyc - 1)
        I, J = np.meshgrid(2 * ic, 2 * jc, indexing="ij")
        corners = ((f[I - 1, J - 1] + f[I - 1, J + 1]) + f[I + 1, J - 1]) + f[I + 1, J + 1]
        edges = ((f[I - 1, J] + f[I + 1, J]) + f[I, J - 1]) + f[I, J + 1]
        c[1:-1, 1:-1] = (1.0 / 16.0 * corners + 1.0 / 8.0 * edges) + 1.0 / 4.0 * f[I, J]
        for cond, rows, cols in ((sides & SIDE_ILO, [0], range(lnyc)), (sides & SIDE_IHI, [lnxc - 1], range(lnyc)),
                                 (sides & SIDE_JLO, range(lnxc), [0]), (sides & SIDE_JHI, range(lnxc), [lnyc - 1])):
            if not cond:
                continue
            for a in rows:
                for b in cols:
                    ghost = (a == 0 and not sides & SIDE_ILO) or (a == lnxc - 1 and not sides & SIDE_IHI) or \
                            (b == 0 and not sides & SIDE_JLO) or (b == lnyc - 1 and not sides & SIDE_JHI)
                    if not ghost:
                        c[a, b] = f[2 * a, 2 * b]

    def prolong_add(self, coarse, fine_u, lnxf, lnyf, lnxc, lnyc, sides):
        e, u = self._v(coarse, lnxc, lnyc).astype(self.comp_dtype), self._v(fine_u, lnxf, lnyf)
        i = np.arange(lnxf)[:, None]; j = np.arange(lnyf)[None, :]
        ic, io, jc, jo = i >> 1, i & 1, j >> 1, j & 1
        valid = (ic + io < lnxc) & (jc + jo < lnyc)
        ic1, jc1 = np.minimum(ic + 1, lnxc - 1), np.minimum(jc + 1, lnyc - 1)
        e00, e01, e10, e11 = e[ic, jc], e[ic, jc1], e[ic1, jc], e[ic1, jc1]
        P = np.where((io == 0) & (jo == 0), e00,
            np.where((io == 1) & (jo == 0), 0.5 * (e00 + e10),
            np.where((io == 0) & (jo == 1), 0.5 * (e00 + e01), 0.25 * (((e00 + e01) + e10) + e11))))
        if sides & SIDE_JHI:
            P = np.where((io == 1) & (jo == 0) & (j == lnyf - 1), 0.0, P)
        if sides & SIDE_IHI:
            P = np.where((io == 0) & (jo == 1) & (i == lnxf - 1), 0.0, P)
        wide = np.result_type(u.dtype, P.dtype)
        u[valid] = (u.astype(wide) + P.astype(wide)).astype(u.dtype)[valid]

    # ---- fused legs with sub-domain semantics (every edge of the local array is treated as fixed) ----------------
    def _prolong_field(self, e, lnxf, lnyf, lnxc, lnyc, ci_off, cj_off, sides, dtype):
        i = np.arange(lnxf)[:, None]; j = np.arange(lnyf)[None, :]
        ic, io, jc, jo = (i >> 1) + ci_off, i & 1, (j >> 1) + cj_off, j & 1
        valid = (ic >= 0) & (jc >= 0) & (ic + io < lnxc) & (jc + jo < lnyc)
        icc, jcc = np.clip(ic, 0, lnxc - 1), np.clip(jc, 0, lnyc - 1)
        ic1, jc1 = np.clip(ic + 1, 0, lnxc - 1), np.clip(jc + 1, 0, lnyc - 1)
        e = e.astype(self.comp_dtype)          # interpolation arithmetic in the grid's dtype (operators/transfer.py:207)
        e00, e01, e10, e11 = e[icc, jcc], e[icc, jc1], e[ic1, jcc], e[ic1, jc1]
        P = np.where((io == 0) & (jo == 0), e00,
            np.where((io == 1) & (jo == 0), 0.5 * (e00 + e10),
            np.where((io == 0) & (jo == 1), 0.5 * (e00 + e01), 0.25 * (((e00 + e01) + e10) + e11))))
        if sides & SIDE_JHI:
            P = np.where((io == 1) & (jo == 0) & (j == lnyf - 1), 0.0, P)
        if sides & SIDE_IHI:
            P = np.where((io == 0) & (jo == 1) & (i == lnxf - 1), 0.0, P)
        return P, valid            # in comp_dtype; the caller adds in the wider of (u, P) and rounds to u's dtype

    def _sweeps(self, sm, v, f, hx, hy, omega, nsweep, poff, a=None):
        """nsweep sweeps of the local array: weighted Jacobi (sm = 0) or red-black GS with the global colouring (sm = 1);
        a: vertex values of the diffusion coefficient (variable-coefficient operator)."""
        if a is not None:
            if sm == 0:
                return O.var_jacobi(v, f, a, hx, hy, omega, nsweep)
            v = v.copy()
            lnx, lny = v.shape
            i = np.arange(1, lnx - 1)[:, None]; j = np.arange(1, lny - 1)[None, :]
            for _ in range(nsweep):
                for colour in (0, 1):
                    m = ((i + j + poff) % 2) == colour
                    upd = O._var_update(v, f, a, hx, hy, omega)
                    inner = v[1:-1, 1:-1]
                    inner[m] = upd[m]
            return v
        if sm == 0:
            return O.jacobi(v, f, hx, hy, omega, nsweep, "vectorized")
        v = v.copy()
        lnx, lny = v.shape
        diag = -2.0 / hx**2 - 2.0 / hy**2
        i = np.arange(1, lnx - 1)[:, None]; j = np.arange(1, lny - 1)[None, :]
        for _ in range(nsweep):
            for colour in (0, 1):
                m = ((i + j + poff) % 2) == colour
                nb = (v[2:, 1:-1] + v[:-2, 1:-1]) / hx**2 + (v[1:-1, 2:] + v[1:-1, :-2]) / hy**2
                upd = (1 - omega) * v[1:-1, 1:-1] + omega * ((f[1:-1, 1:-1] + nb) / (-diag))
                inner = v[1:-1, 1:-1]
                inner[m] = upd[m]
        return v

    def var_rdiag(self, a, rd, lnx, lny, hx, hy, sigma=0.0):
        pass                      # the oracle's smoothers form 1 / D themselves (oracle/mg_oracle.py: _var_update)

    def down_leg(self, sm, u, rhs, out, rhs_c, lnx, lny, lnxc, lnyc, ci_off, cj_off, hx, hy, omega, coeff, nsweep, zero_init, poff,
                 select=0, inner=None, acoef=None, rdiag=None):
        assert select == 0          # no streams on the CPU: the driver never splits the launch here
        f = self._v(rhs, lnx, lny)
        a = None if acoef is None else self._v(acoef, lnx, lny)
        v = np.zeros_like(f) if zero_init else self._v(u, lnx, lny).copy()
        v = self._sweeps(sm, v, f, hx, hy, omega, nsweep, poff, a)
        self._v(out, lnx, lny)[1:, :] = v[1:, :]                    # the kernel never writes row 0
        r = O.residual(v, f, hx, hy, coeff) if a is None else O.var_residual(v, f, a, hx, hy, coeff)
        c = self._v(rhs_c, lnxc, lnyc)
        ic = np.arange(1, lnxc - 1); jc = np.arange(1, lnyc - 1)
        fi = 2 * (ic - ci_off); fj = 2 * (jc - cj_off)
        oki = (fi >= 1) & (fi <= lnx - 2); okj = (fj >= 1) & (fj <= lny - 2)
        I, J = np.meshgrid(fi[oki], fj[okj], indexing="ij")
        corners = ((r[I - 1, J - 1] + r[I - 1, J + 1]) + r[I + 1, J - 1]) + r[I + 1, J + 1]
        edges = ((r[I - 1, J] + r[I + 1, J]) + r[I, J - 1]) + r[I, J + 1]
        c[np.ix_(ic[oki], jc[okj])] = (1.0 / 16.0 * corners + 1.0 / 8.0 * edges) + 1.0 / 4.0 * r[I, J]

    def up_leg(self, sm, u, rhs, out, e_c, lnx, lny, lnxc, lnyc, ci_off, cj_off, sides, hx, hy, omega, coeff, nsweep, poff, window=None,
               acoef=None, rdiag=None):
        f = self._v(rhs, lnx, lny)
        a = None if acoef is None else self._v(acoef, lnx, lny)
        v = self._v(u, lnx, lny).copy()
        P, valid = self._prolong_field(self._v(e_c, lnxc, lnyc), lnx, lny, lnxc, lnyc, ci_off, cj_off, sides, v.dtype)
        wide = np.result_type(v.dtype, P.dtype)
        v[valid] = (v.astype(wide) + P.astype(wide)).astype(v.dtype)[valid]
        v = self._sweeps(sm, v, f, hx, hy, omega, nsweep, poff, a)
        self._v(out, lnx, lny)[1:, :] = v[1:, :]
        if window is None:
            return None
        r = O.residual(v, f, hx, hy, coeff) if a is None else O.var_residual(v, f, a, hx, hy, coeff)
        i_lo, i_hi, j_lo, j_hi = window
        w = r[max(i_lo, 1):min(i_hi, lnx - 1), max(j_lo, 1):min(j_hi, lny - 1)].astype(np.float64)
        return torch.tensor([float(np.sum(w * w))], dtype=torch.float64)

    span_min_cells = 1100 * 1100          # as the library: blocks above ~1100^2 cells (tests lower it to reach the path on small grids)

    def span_ok(self, sm, u, e_c, lnx, lny):
        return sm == 0 and u.dtype == e_c.dtype and lnx * lny > self.span_min_cells

    def span_leg(self, sm, u, rhs, out_mid, out_next, e_c, rhs_c, lnx, lny, lnxc, lnyc, ci_off, cj_off, sides, hx, hy, omega, coeff,
                 nsweep_post, nsweep_pre, poff, window):
        """the library's spanning leg = its up leg followed by its down leg (same bits)"""
        res = self.up_leg(sm, u, rhs, out_mid, e_c, lnx, lny, lnxc, lnyc, ci_off, cj_off, sides, hx, hy, omega, coeff, nsweep_post, poff, window)
        self.down_leg(sm, out_mid, rhs, out_next, rhs_c, lnx, lny, lnxc, lnyc, ci_off, cj_off, hx, hy, omega, coeff, nsweep_pre, False, poff)
        return res

    def inject_ring(self, fine, coarse, lnxf, lnyf, lnxc, lnyc, sides, ci_off, cj_off):
        f, c = self._v(fine, lnxf, lnyf), self._v(coarse, lnxc, lnyc)
        for side, rows, cols in ((SIDE_ILO, [0], range(lnyc)), (SIDE_IHI, [lnxc - 1], range(lnyc)),
                                 (SIDE_JLO, range(lnxc), [0]), (SIDE_JHI, range(lnxc), [lnyc - 1])):
            if not sides & side:
                continue
            for a in rows:
                for b in cols:
                    fi, fj = 2 * (a - ci_off), 2 * (b - cj_off)
                    if 0 <= fi < lnxf and 0 <= fj < lnyf:
                        c[a, b] = f[fi, fj]

    # replicated coarse hierarchy: the oracle's single-domain cycle
    def coarse_setup(self, NX, NY, domain, cfg):
        self._kind = {0: "jacobi", 1: "rbgs"}[cfg["smoother"]]
        self._cfg, self._domain = dict(cfg), domain
        self._shape = (NX, NY)
        self._pm = None
        if self.mixed:      # the replicated levels keep the GLOBAL split: level >= mixed_split (counted from here) is fp32
            split = int(cfg.get("mixed_split", 0))

            class _Split:
                def for_level(self_, level, nlevels):
                    return "float32" if level >= split else "float64"
                convert = staticmethod(lambda arr, p: arr if arr.dtype == np.dtype(p) else arr.astype(p))
            self._pm = _Split()
        self._build_mg(None)

    def _build_mg(self, a):
        cfg = self._cfg
        NX, NY = self._shape
        dt = np.float64 if self.mixed else self.np_dtype
        args = (self._domain, dt, cfg["coeff"], cfg["levels"], cfg["cycle"], cfg["pre"], cfg["post"], self._kind, cfg["omega"],
                "vectorized", cfg["coarse_tol"], cfg["coarse_maxit"])
        self._mg = O.MGOracle(NX, NY, *args) if a is None else O.VarMGOracle(a, *args)

    def coarse_coefficient(self, a_host):
        self._build_mg(None if a_host is None else np.asarray(a_host, dtype=np.float64))

    def coarse_begin(self, rhs_global):
        self._mg.rhs[0] = self._v(rhs_global, *self._shape).copy()
        self._e = np.zeros(self._shape, dtype=self._mg.rhs[0].dtype)

    def coarse_cycle(self):
        self._e = self._mg.cycle_once(self._e, 0, self._pm)

    def coarse_end(self, out_global):
        self._v(out_global, *self._shape)[...] = self._e

    def close(self):
        pass


def assemble(solver, NX, NY, dtype=np.float64):
    """Global solution from the exclusive windows of the ranks in this process."""
    out = np.full((NX, NY), np.nan, dtype=dtype)
    for r in solver.ranks:
        b, u = solver.local_solution(r)
        out[b.gx0 + b.i_lo:b.gx0 + b.i_hi, b.gy0 + b.j_lo:b.gy0 + b.j_hi] = u[b.i_lo:b.i_hi, b.j_lo:b.j_hi]
    return out
