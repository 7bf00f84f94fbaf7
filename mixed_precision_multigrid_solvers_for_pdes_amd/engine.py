"""Thin object wrapper of one mg_handle (include/mghip.h): the device-resident hierarchy."""
import ctypes as C

import numpy as np

from . import _lib


def _direct_code(coarse_direct):
    """mg_config.coarse_direct for the engine's keyword (see MultigridEngine)"""
    import os
    if coarse_direct is None:
        coarse_direct = {"0": False, "1": True}.get(os.environ.get("MG_COARSE_DIRECT", "auto"), "auto")
    if isinstance(coarse_direct, str):
        if coarse_direct != "auto":
            raise ValueError(f"coarse_direct: True, False, 'auto' or None, not {coarse_direct!r}")
        return -1
    return int(bool(coarse_direct))


class MultigridEngine:
    """Owns a device-resident multigrid hierarchy.  All arguments are the fields of mg_config."""

    def __init__(self, nx, ny, domain=(0.0, 1.0, 0.0, 1.0), coeff=-1.0, max_levels=4, cycle="V", pre=2, post=2,
                 smoother=_lib.MG_JACOBI, omega=0.8, coarse_tol=1e-12, coarse_maxit=1000,
                 precision=_lib.MG_PREC_DOUBLE, switch_threshold=1e-6, memory_threshold_gb=4.0,
                 adaptive_reference_rule=False, device=0, profile=False, colour_offset=0, fused=2, tail=True, speculate=True, fmg_cycles=0,
                 mixed_split=0, coarse_direct=None):
        lib = _lib.load()
        # coarse_direct: True the nine unknowns of a 5 x 5 coarsest grid are solved directly (u = A^-1 f; within 1e-12 of the
        # reference's iterates, not bit-identical), False by the reference's Gauss-Seidel iteration to coarse_tol (bit-identical),
        # "auto" = True (W- / F-cycles visit the coarsest grid 2^(L-1) times per cycle and spent most of their time in that
        # iteration; a V-cycle saves its ~20 sweeps); None (default): what the environment variable MG_COARSE_DIRECT says
        # ("0" / "1" / "auto"), "auto" without it -- the test suite pins "0": its comparisons with the oracle and between
        # engine paths are bit for bit.
        # tail: True / 1 the coarse levels run in the register-resident one-workgroup kernel where it applies (dyadic square
        # levels <= 65^2, csrc/mg_tail_kernels.hpp) and in the LDS one elsewhere; 2 the LDS kernel only; False / 0 one launch pair per level.
        # fused: 0 / False one launch per operator; 1 / True fused legs tiled through LDS; 2 (default) the same legs
        # register-blocked on levels above ~1100^2 cells; 3 register-blocked on every level (include/mghip.h mg_config.fused)
        if isinstance(cycle, str):
            if cycle not in _lib.CYCLES:
                raise ValueError(f"unknown cycle type {cycle!r}")
            cycle = _lib.CYCLES[cycle]
        cfg = _lib.MgConfig(int(nx), int(ny), float(domain[0]), float(domain[1]), float(domain[2]), float(domain[3]),
                            float(coeff), int(max_levels), int(cycle), int(pre), int(post), int(smoother),
                            float(omega), float(coarse_tol), int(coarse_maxit), int(precision),
                            float(switch_threshold), float(memory_threshold_gb), int(bool(adaptive_reference_rule)),
                            int(device), int(bool(profile)), int(colour_offset), int(fused), int(tail), int(fmg_cycles), (2 if speculate is True else int(speculate)),
                            _direct_code(coarse_direct), int(mixed_split))
        self.cfg = cfg
        self._h = C.c_void_p(None)
        _lib.check(lib.mg_create(C.byref(cfg), C.byref(self._h)))
        self._lib = lib
        n = C.c_int(0)
        self._check(lib.mg_num_levels(self._h, C.byref(n)))
        self.num_levels = n.value
        self.shapes = []
        for l in range(self.num_levels):
            a, b = C.c_int(0), C.c_int(0)
            self._check(lib.mg_level_shape(self._h, l, C.byref(a), C.byref(b)))
            self.shapes.append((a.value, b.value))
        self.nx, self.ny = int(nx), int(ny)

    def _check(self, rc):
        _lib.check(rc, self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mg_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- whole solve ---------------------------------------------------------------------
    def solve(self, rhs, u0=None, tol=1e-8, max_iterations=50, out_dtype=None):
        rhs = _lib.as_c(rhs)
        if rhs.shape != (self.nx, self.ny):
            raise ValueError("Multigrid not properly setup or grid mismatch")
        dt = rhs.dtype if out_dtype is None else np.dtype(out_dtype)
        rhs = np.ascontiguousarray(rhs, dtype=dt)
        u0c = None if u0 is None else np.ascontiguousarray(u0, dtype=dt)
        if u0c is not None and u0c.shape != rhs.shape:
            raise ValueError("initial guess shape does not match the grid")
        out = np.empty_like(rhs)
        hist = (C.c_double * max_iterations)()
        prec = (C.c_int32 * max_iterations)()
        nit, conv = C.c_int(0), C.c_int(0)
        stats = _lib.MgStats()
        self._check(self._lib.mg_solve(self._h, _lib.ptr(rhs), None if u0c is None else _lib.ptr(u0c), _lib.ptr(out),
                                       _lib.dtype_code(dt), float(tol), int(max_iterations), hist, max_iterations,
                                       C.byref(nit), C.byref(conv), prec, C.byref(stats)))
        n = nit.value
        return out, {
            "iterations": n, "converged": bool(conv.value),
            "residual_history": [hist[i] for i in range(n)],
            "precision_codes": [prec[i] for i in range(n)],
            "initial_residual": stats.initial_residual,
            "solve_seconds": stats.solve_seconds, "h2d_seconds": stats.h2d_seconds,
            "d2h_seconds": stats.d2h_seconds, "precision_switches": stats.precision_switches,
            "switch_reason": _lib.SWITCH_REASONS.get(stats.switch_reason), "fp32_floor": stats.fp32_floor,
            "last_coarse_sweeps": stats.last_coarse_sweeps,
        }

    def iterate(self, tol=0.0, max_iterations=20):
        """The policy + cycle + norm loop on the fields already resident on the device (no transfers)."""
        hist = (C.c_double * max_iterations)()
        prec = (C.c_int32 * max_iterations)()
        nit, conv = C.c_int(0), C.c_int(0)
        stats = _lib.MgStats()
        self._check(self._lib.mg_iterate(self._h, float(tol), int(max_iterations), hist, max_iterations,
                                         C.byref(nit), C.byref(conv), prec, C.byref(stats)))
        n = nit.value
        return {"iterations": n, "converged": bool(conv.value), "residual_history": [hist[i] for i in range(n)],
                "precision_codes": [prec[i] for i in range(n)], "initial_residual": stats.initial_residual,
                "solve_seconds": stats.solve_seconds, "precision_switches": stats.precision_switches,
                "switch_reason": _lib.SWITCH_REASONS.get(stats.switch_reason), "fp32_floor": stats.fp32_floor}

    def set_coefficient(self, a):
        """Variable-coefficient operator A = coeff * div(a grad .): vertex values of a on the fine grid (None: back
        to the constant-coefficient operator)."""
        if a is None:
            self._check(self._lib.mg_set_coefficient(self._h, None, _lib.MG_F64))
            return
        a = _lib.as_c(a)
        if a.shape != (self.nx, self.ny):
            raise ValueError(f"coefficient shape {a.shape} doesn't match grid shape {(self.nx, self.ny)}")
        self._check(self._lib.mg_set_coefficient(self._h, _lib.ptr(a), _lib.dtype_code(a.dtype)))

    def set_shift(self, sigma):
        """Helmholtz shift: every level's operator becomes coeff * (Laplacian - sigma I) (include/mghip.h: mg_set_shift)."""
        self._check(self._lib.mg_set_shift(self._h, float(sigma)))

    # ---- device-resident stepping ---------------------------------------------------------
    def set_rhs(self, rhs):
        rhs = _lib.as_c(rhs)
        if rhs.shape != (self.nx, self.ny):
            raise ValueError("Multigrid not properly setup or grid mismatch")
        self._check(self._lib.mg_set_rhs(self._h, _lib.ptr(rhs), _lib.dtype_code(rhs.dtype)))

    def set_solution(self, u0=None, dtype=np.float64):
        if u0 is None:
            self._check(self._lib.mg_set_solution(self._h, None, _lib.dtype_code(dtype)))
        else:
            u0 = _lib.as_c(u0)
            self._check(self._lib.mg_set_solution(self._h, _lib.ptr(u0), _lib.dtype_code(u0.dtype)))

    def get_solution(self, dtype=np.float64):
        out = np.empty((self.nx, self.ny), dtype=dtype)
        self._check(self._lib.mg_get_solution(self._h, _lib.ptr(out), _lib.dtype_code(dtype)))
        return out

    def cycle(self, n=1):
        self._check(self._lib.mg_cycle(self._h, int(n)))

    def fmg(self, cycles_per_level=1):
        """Full-multigrid initial guess from the resident rhs (the boundary ring of the current iterate is kept)."""
        self._check(self._lib.mg_fmg(self._h, int(cycles_per_level)))

    def residual_norm(self):
        out = C.c_double(0.0)
        self._check(self._lib.mg_residual_norm(self._h, C.byref(out)))
        return out.value

    def set_working_precision(self, dtype):
        self._check(self._lib.mg_set_working_precision(self._h, _lib.dtype_code(dtype)))

    def synchronize(self):
        self._check(self._lib.mg_synchronize(self._h))

    def time_op(self, op, level=0, dtype=np.float64, reps=20):
        ops = {"jacobi": 0, "rbgs": 1, "residual": 2, "residual_norm": 3, "restrict": 4, "prolong": 5, "cycle": 6,
               "down_leg": 7, "up_leg": 8, "sweeps2": 9, "jacobi_hbm": 10, "stream_hbm": 11, "span_leg": 12, "span_leg_nomid": 13}
        out = C.c_double(0.0)
        self._check(self._lib.mg_time_op(self._h, ops[op] if isinstance(op, str) else int(op), int(level),
                                         _lib.dtype_code(dtype), int(reps), C.byref(out)))
        return out.value

    def level_timings(self):
        res = {}
        for l in range(self.num_levels):
            t = (C.c_double * 3)()
            self._check(self._lib.mg_level_timings(self._h, l, t))
            res[l] = {"smooth_time": t[0], "restrict_time": t[1], "prolong_time": t[2]}
        return res
