"""Block domain decomposition of the multigrid hierarchy over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the
tests).  The reference has no communication backend at all (gpu/multi_gpu.py:540-607 solves
sub-domains independently, gpu/multi_gpu_solver.py:90-185 copies slices between CuPy devices); what is
kept from it is the partitioning idea (2-D blocks with a 1-cell overlap, gpu/multi_gpu.py:386-476,
gpu/multi_gpu_solver.py:30-64) -- the algorithm here is the SAME V/W-cycle as the single-GPU engine,
decomposition-invariant by construction:

  * the global vertex grid (NX, NY) is cut at indices c_k = k (NX-1)/px (even on every distributed
    level); rank (rx, ry) stores global rows c_rx .. c_rx+1 + 1: its owned cells plus a 1-cell ring
    that is either the physical boundary or a ghost copy of the neighbour's edge.  Local cell (0,0) has
    an even global index, so coarse cell (ic, jc) sits on local fine cell (2ic, 2jc) on every rank and
    the red/black colouring is the global one;
  * mode "fused" (default; communication-avoiding): every rank keeps a ghost ZONE of G cells and runs the same
    two fused kernels per level as the single-GPU engine (down leg: 2 sweeps + residual + restriction; up leg:
    prolongation + 2 sweeps [+ norm]) on its whole local array, recomputing inside the ghost zone what the
    neighbour computes too; each Jacobi sweep / GS colour pass / residual / transfer invalidates one more ghost
    cell from the outside; G = 7 (weighted Jacobi) and G = 13 (red-black GS) are the smallest odd widths for
    which the owned cells stay exact through every V / W / F visit (GHOST_FUSED below).
    Exchanges per cycle: the fine iterate once (G rows / columns per neighbour) and each coarse right-hand side
    once -- L_d + 1 exchanges instead of 5 L_d, each a few hundred KB instead of 16 KB, and two launches per level;
  * mode "per_operator": 1-cell ghost ring, one launch per operator (the kernels pass the ring through); every
    sweep (every colour) is followed by a halo exchange of u; the residual gets one exchange (with corners)
    before full-weighting restriction; prolongation interpolates the ghost ring from the exchanged coarse ghost
    values, so no exchange is needed after the correction;
  * below `agglomerate_at` points per direction the remaining coarse hierarchy is solved redundantly on
    every GPU by the single-GPU engine after one all-gather of the coarse right-hand side: no broadcast
    back, no latency-bound tiny halo messages;
  * ||r|| is an all-reduce of one fp64 partial sum per rank over exactly the cells each rank owns.

Messages are G rows / columns (118 KB at 4097^2 fp32, G = 7; 16-32 KB per 1-cell ring in per-operator mode):
latency-bound, each neighbour pair on its own xGMI link.  Fields live in torch tensors (device memory, streams); the arithmetic is libmghip's
device-pointer entry points (mg_dev_*).  `ops` and `comm` are injected so that the decomposition logic
runs unchanged (a) on CPU under gloo with a NumPy stand-in for the kernels (tests) and (b) with several
virtual ranks in one process on one GPU (tests), besides (c) the real thing.
"""
import ctypes as C
import json
import math
import os
import time

import numpy as np

from . import _lib

SIDE_ILO, SIDE_IHI, SIDE_JLO, SIDE_JHI = 1, 2, 4, 8


# ------------------------------------------------------------------------------------------------
# index bookkeeping (pure Python, no device)
# ------------------------------------------------------------------------------------------------
def hierarchy_shapes(nx, ny, max_levels):
    """Global level shapes by the reference's rule (solvers/multigrid.py:153-171)."""
    shapes = [(nx, ny)]
    for _ in range(1, max_levels):
        a, b = shapes[-1]
        if (a - 1) % 2 or (b - 1) % 2:
            break
        c = ((a - 1) // 2 + 1, (b - 1) // 2 + 1)
        if c[0] < 5 or c[1] < 5:
            break
        shapes.append(c)
    return shapes


def process_grid(world):
    """px x py with px >= py, as square as possible (1, 2x1, 2x2, 4x2)."""
    py = int(math.sqrt(world))
    while world % py:
        py -= 1
    return world // py, py


class Block:
    """One rank's block of one level: owned cells, a ghost zone of `G` cells towards every neighbour, the physical
    boundary row/column where the block touches the domain boundary.  Local (0, 0) has an even global index."""

    def __init__(self, NX, NY, px, py, rx, ry, G=1):
        mx, my = (NX - 1) // px, (NY - 1) // py
        self.NX, self.NY, self.G = NX, NY, G
        self.gx0 = 0 if rx == 0 else rx * mx - (G - 1)              # global index of local (0, 0)
        self.gy0 = 0 if ry == 0 else ry * my - (G - 1)
        gx1 = NX - 1 if rx == px - 1 else (rx + 1) * mx + G          # global index of the last local row
        gy1 = NY - 1 if ry == py - 1 else (ry + 1) * my + G
        self.lnx, self.lny = gx1 - self.gx0 + 1, gy1 - self.gy0 + 1
        self.sides = ((SIDE_ILO if rx == 0 else 0) | (SIDE_IHI if rx == px - 1 else 0) |
                      (SIDE_JLO if ry == 0 else 0) | (SIDE_JHI if ry == py - 1 else 0))
        # owned interior cells (local indices, inclusive)
        self.oi_lo = rx * mx + 1 - self.gx0
        self.oi_hi = (NX - 2 if rx == px - 1 else (rx + 1) * mx) - self.gx0
        self.oj_lo = ry * my + 1 - self.gy0
        self.oj_hi = (NY - 2 if ry == py - 1 else (ry + 1) * my) - self.gy0
        # exclusive window: owned cells plus the adjacent physical boundary cells (a disjoint cover of the grid)
        self.i_lo = 0 if rx == 0 else self.oi_lo
        self.i_hi = self.lnx if rx == px - 1 else self.oi_hi + 1
        self.j_lo = 0 if ry == 0 else self.oj_lo
        self.j_hi = self.lny if ry == py - 1 else self.oj_hi + 1

    def coarse_offsets(self, coarse):
        """(ci_off, cj_off): coarse local (ic, jc) sits on fine local (2 (ic - ci_off), 2 (jc - cj_off))."""
        return (self.gx0 - 2 * coarse.gx0) // 2, (self.gy0 - 2 * coarse.gy0) // 2


def distributed_levels(shapes, px, py, agglomerate_at, G=1):
    """Number of leading levels that stay distributed.  A level is distributed while its cuts are even
    (so the next level lines up), its blocks own at least max(4, G + 1) rows/cols -- and those of the level below
    at least G, whose ghost zone the correction is cut out for -- and it is larger than `agglomerate_at` points in
    some direction; at least one level is always left for the replicated part."""
    n = 0
    need = max(4, G + 1)
    for (NX, NY) in shapes[:-1]:
        if (NX - 1) % px or (NY - 1) % py:
            break
        mx, my = (NX - 1) // px, (NY - 1) // py
        if (px > 1 and (mx % 2 or mx < need or mx // 2 < G)) or (py > 1 and (my % 2 or my < need or my // 2 < G)):
            break
        if max(NX, NY) <= agglomerate_at:
            break
        n += 1
    return n


# ------------------------------------------------------------------------------------------------
# kernels on torch tensors (device pointers into libmghip)
# ------------------------------------------------------------------------------------------------
class HipOps:
    """mg_dev_* on CUDA tensors.  A field is a 2-D tensor (lnx, ld) whose first lny columns are the data; its precision
    is the tensor's dtype (levels of one hierarchy may differ: per-level mixed precision)."""

    def __init__(self, dtype, device, managed_single=False, mixed=False):
        """dtype: the default field precision (`alloc` without a dtype).
        managed_single (fp32 fields only): the reference's PrecisionManager('single') layout on a float64 Grid --
        interpolation in fp64 and the coarsest level solved in fp64 (otherwise an fp32 coarsest solve can never
        meet the 1e-12 tolerance and burns its 1000 sweeps on every visit, exactly like Grid(dtype=float32) does).
        mixed: PrecisionManager('mixed') on a float64 Grid (core/precision.py:337-357): the caller allocates coarse
        levels in fp32; interpolation runs in fp64 (the grid dtype)."""
        import torch
        self.torch = torch
        self.lib = _lib.load()
        self.np_dtype = np.dtype(dtype)
        self.dt = _lib.dtype_code(dtype)
        self.managed = bool(managed_single) and self.dt == _lib.MG_F32
        self.mixed = bool(mixed)
        self.comp_dt = _lib.MG_F64 if (self.managed or self.mixed) else self.dt
        self.tdtype = torch.float32 if self.dt == _lib.MG_F32 else torch.float64
        self.device = device
        self.scratch = torch.zeros(2048, dtype=torch.float64, device=device)          # grown per field shape (_scratch_for)
        self.acc = torch.zeros(1, dtype=torch.float64, device=device)
        self._engine = None
        self._coarse_ring_valid = False # the replicated engine holds the boundary ring of the current problem's coarse rhs
        self.rec = None                 # dist_plan.PlanRecorder while a cycle is being recorded

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def _scratch_for(self, lnx, lny):
        """Partial-sum scratch of at least mg_dev_scratch_bytes(lnx, lny) (the library writes one fp64 per workgroup)."""
        nbytes = C.c_int64(0)
        _lib.check(self.lib.mg_dev_scratch_bytes(int(lnx), int(lny), C.byref(nbytes)))
        if self.scratch.numel() * 8 < nbytes.value:
            self.scratch = self.torch.zeros(nbytes.value // 8, dtype=self.torch.float64, device=self.device)
        return self._p(self.scratch)

    def _code(self, t):
        return _lib.MG_F32 if t.dtype == self.torch.float32 else _lib.MG_F64

    def alloc(self, lnx, lny, dtype=None):
        code = self.dt if dtype is None else _lib.dtype_code(dtype)
        ld = C.c_int(0)
        _lib.check(self.lib.mg_pitch_elems(code, lny, C.byref(ld)))
        return self.torch.zeros((lnx, ld.value), dtype=self.torch.float32 if code == _lib.MG_F32 else self.torch.float64,
                                device=self.device)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def jacobi(self, u, rhs, out, lnx, lny, hx, hy, omega):
        _lib.check(self.lib.mg_dev_jacobi(self._code(u), lnx, lny, u.stride(0), hx, hy, omega, self._p(u), self._p(rhs),
                                          self._p(out), self._stream()))

    def rbgs_colour(self, u, rhs, lnx, lny, hx, hy, omega, colour, offset):
        _lib.check(self.lib.mg_dev_rbgs_colour(self._code(u), lnx, lny, u.stride(0), hx, hy, omega, colour, offset,
                                               self._p(u), self._p(rhs), self._stream()))

    def residual(self, u, f, r, lnx, lny, hx, hy, coeff):
        _lib.check(self.lib.mg_dev_residual(self._code(u), lnx, lny, u.stride(0), hx, hy, coeff, self._p(u), self._p(f),
                                            self._p(r), self._stream()))

    def sumsq(self, field, i_lo, i_hi, j_lo, j_hi):
        """fp64 sum of squares of the window as a 1-element device tensor."""
        _lib.check(self.lib.mg_dev_sumsq(self._code(field), field.stride(0), i_lo, i_hi, j_lo, j_hi, self._p(field),
                                         self._scratch_for(i_hi, j_hi), self._p(self.acc), self._stream()))
        return self.acc.clone()

    def restrict(self, fine, coarse, lnxf, lnyf, lnxc, lnyc, sides):
        _lib.check(self.lib.mg_dev_restrict_fw(self._code(fine), self._code(coarse), lnxf, lnyf, fine.stride(0), lnxc, lnyc,
                                               coarse.stride(0), sides, self._p(fine), self._p(coarse), self._stream()))

    def prolong_add(self, coarse, fine_u, lnxf, lnyf, lnxc, lnyc, sides):
        _lib.check(self.lib.mg_dev_prolong_add(self._code(coarse), self._code(fine_u), self.comp_dt, lnxf, lnyf, fine_u.stride(0),
                                               lnxc, lnyc, coarse.stride(0), sides, self._p(coarse), self._p(fine_u), self._stream()))

    # fused legs (mode "fused"): the single-GPU engine's kernels on the local array with its ghost zone
    supports_overlap = True
    plan_capable = True               # cycles can be recorded into a native plan (dist_plan.py)

    def var_rdiag(self, a, rd, lnx, lny, hx, hy, sigma=0.0):
        """reciprocal diagonal of -div(a grad .) on this array (mg_dev_var_rdiag): what the variable-coefficient sweeps multiply by"""
        _lib.check(self.lib.mg_dev_var_rdiag(self._code(a), lnx, lny, a.stride(0), hx, hy, float(sigma), self._p(a), self._p(rd), self._stream()))

    def down_leg(self, sm, u, rhs, out, rhs_c, lnx, lny, lnxc, lnyc, ci_off, cj_off, hx, hy, omega, coeff, nsweep, zero_init, poff,
                 select=0, inner=None, acoef=None, rdiag=None):
        """select: 0 all tiles; 1 only tiles that need nothing outside `inner` = (i_lo, i_hi, j_lo, j_hi); 2 the others.
        acoef / rdiag: vertex values of the diffusion coefficient on this array and its reciprocal diagonal (var_rdiag);
        None: constant-coefficient operator."""
        rect = (C.c_int * 4)(*inner) if inner is not None else None
        _lib.check(self.lib.mg_dev_down_leg_var(sm, self._code(rhs), self._code(rhs_c), lnx, lny, rhs.stride(0), lnxc, lnyc,
                                                rhs_c.stride(0), ci_off, cj_off, hx, hy, omega, coeff, nsweep, int(zero_init), poff,
                                                None if zero_init else self._p(u), self._p(rhs), self._p(out), self._p(rhs_c),
                                                self._stream(), int(select), rect, None if acoef is None else self._p(acoef),
                                                None if rdiag is None else self._p(rdiag)))
        if self.rec is not None:
            clamp = lambda v: max(-(1 << 30), min(1 << 30, int(v)))
            self.rec.emit(_lib.MG_PLAN_DOWN_LEG,
                          i=(sm, self._code(rhs), self._code(rhs_c), lnx, lny, rhs.stride(0), lnxc, lnyc, rhs_c.stride(0), ci_off, cj_off,
                             nsweep, int(zero_init), poff, int(select), int(inner is not None)) + tuple(clamp(v) for v in (inner or (0, 0, 0, 0))),
                          d=(hx, hy, omega, coeff), p=(None if zero_init else u, rhs, out, rhs_c, acoef, rdiag))

    def up_leg(self, sm, u, rhs, out, e_c, lnx, lny, lnxc, lnyc, ci_off, cj_off, sides, hx, hy, omega, coeff, nsweep, poff,
               window=None, acoef=None, rdiag=None):
        """out = sweeps(u + P e_c); with `window` = (i_lo, i_hi, j_lo, j_hi) also returns sum r^2 over it (device tensor)."""
        w = window or (0, 0, 0, 0)
        res = self.torch.empty(1, dtype=self.torch.float64, device=self.device) if window is not None else self.acc
        self._scratch_for(lnx, lny)
        _lib.check(self.lib.mg_dev_up_leg_var(sm, self._code(u), self._code(e_c), self.comp_dt, lnx, lny, u.stride(0), lnxc, lnyc,
                                              e_c.stride(0), ci_off, cj_off, sides, hx, hy, omega, coeff, nsweep, poff, self._p(u),
                                              self._p(rhs), self._p(out), self._p(e_c), int(window is not None), w[0], w[1], w[2],
                                              w[3], self._p(self.scratch), self._p(res), self._stream(),
                                              None if acoef is None else self._p(acoef), None if rdiag is None else self._p(rdiag)))
        if self.rec is not None:
            self.rec.emit(_lib.MG_PLAN_UP_LEG,
                          i=(sm, self._code(u), self._code(e_c), self.comp_dt, lnx, lny, u.stride(0), lnxc, lnyc, e_c.stride(0), ci_off,
                             cj_off, sides, nsweep, poff, int(window is not None)) + tuple(w),
                          d=(hx, hy, omega, coeff), p=(u, rhs, out, e_c, self.scratch, res, acoef, rdiag))
        return res if window is not None else None

    def span_ok(self, sm, u, e_c, lnx, lny):
        """the spanning leg serves this block (weighted Jacobi, one dtype, above ~1100^2 cells: include/mghip.h)"""
        return bool(self.lib.mg_dev_span_leg_ok(sm, self._code(u), self._code(e_c), self.comp_dt, lnx, lny))

    def span_leg(self, sm, u, rhs, out_mid, out_next, e_c, rhs_c, lnx, lny, lnxc, lnyc, ci_off, cj_off, sides, hx, hy, omega, coeff,
                 nsweep_post, nsweep_pre, poff, window):
        """up_leg of cycle k (u -> out_mid, sum r^2 over `window`) and down_leg of cycle k + 1 (-> out_next, rhs_c) in one
        launch (mg_dev_span_leg); returns the sum as a device tensor"""
        res = self.torch.empty(1, dtype=self.torch.float64, device=self.device)
        self._scratch_for(lnx, lny)
        w = window
        _lib.check(self.lib.mg_dev_span_leg(sm, self._code(u), self._code(e_c), self.comp_dt, lnx, lny, u.stride(0), lnxc, lnyc,
                                            e_c.stride(0), ci_off, cj_off, sides, hx, hy, omega, coeff, nsweep_post, nsweep_pre, poff,
                                            self._p(u), self._p(rhs), self._p(out_mid), self._p(out_next), self._p(e_c), self._p(rhs_c),
                                            w[0], w[1], w[2], w[3], self._p(self.scratch), self._p(res), self._stream()))
        if self.rec is not None:
            self.rec.emit(_lib.MG_PLAN_SPAN_LEG,
                          i=(sm, self._code(u), self._code(e_c), self.comp_dt, lnx, lny, u.stride(0), lnxc, lnyc, e_c.stride(0), ci_off,
                             cj_off, sides, nsweep_post, nsweep_pre, poff) + tuple(w),
                          d=(hx, hy, omega, coeff), p=(u, rhs, out_mid, out_next, e_c, rhs_c, self.scratch, res))
        return res

    def inject_ring(self, fine, coarse, lnxf, lnyf, lnxc, lnyc, sides, ci_off, cj_off):
        _lib.check(self.lib.mg_dev_inject_ring(self._code(fine), self._code(coarse), lnxf, lnyf, fine.stride(0), lnxc, lnyc,
                                               coarse.stride(0), sides, ci_off, cj_off, self._p(fine), self._p(coarse), self._stream()))

    # replicated coarse hierarchy = the single-GPU engine on this GPU, queued on the same stream
    def coarse_setup(self, NX, NY, domain, cfg):
        """cfg["mixed_split"] (per-level mixed only): first fp32 level counted from the agglomeration level; <= 0 means
        every level of the replicated part but the coarsest is fp32."""
        from .engine import MultigridEngine
        split = 0
        if self.mixed:
            split = int(cfg.get("mixed_split", 0))
            prec = _lib.MG_PREC_MIXED_LEVELS if split > 0 else _lib.MG_PREC_SINGLE_MANAGED
        elif self.dt == _lib.MG_F32:
            prec = _lib.MG_PREC_SINGLE_MANAGED if self.managed else _lib.MG_PREC_SINGLE
        else:
            prec = _lib.MG_PREC_DOUBLE
        self._engine = MultigridEngine(NX, NY, domain, cfg["coeff"], cfg["levels"], cfg["cycle"], cfg["pre"], cfg["post"],
                                       cfg["smoother"], cfg["omega"], cfg["coarse_tol"], cfg["coarse_maxit"], prec,
                                       device=self.device.index or 0, mixed_split=max(split, 0))
        _lib.check(self.lib.mg_set_stream(self._engine._h, self._stream(), 0))

    def coarse_coefficient(self, a_host):
        """vertex values of the diffusion coefficient on the agglomeration level (host array; None: constant)"""
        self._engine.set_coefficient(a_host)

    def coarse_begin(self, rhs_global, same_ring=False):
        """same_ring: the boundary ring of rhs_global equals that of the previous call (the coarse right-hand side of a
        decomposed cycle: its ring is the injected ring of f, the same cycle after cycle) -- the replicated engine keeps the
        rings of its coarser levels instead of injecting them again (mg_update_rhs_device)."""
        e = self._engine
        same_ring = bool(same_ring) and self._coarse_ring_valid
        _lib.check(self.lib.mg_set_stream(e._h, self._stream(), 0))
        fn = self.lib.mg_update_rhs_device if same_ring else self.lib.mg_set_rhs_device
        _lib.check(fn(e._h, self._p(rhs_global), rhs_global.stride(0), self._code(rhs_global)))
        self._coarse_ring_valid = True
        _lib.check(self.lib.mg_zero_solution_device(e._h))
        if self.rec is not None:
            self.rec.emit(_lib.MG_PLAN_COARSE_BEGIN, i=(rhs_global.stride(0), self._code(rhs_global), int(same_ring)), p=(e._h.value, rhs_global))

    def coarse_cycle(self):
        self._engine.cycle(1)
        if self.rec is not None:
            self.rec.emit(_lib.MG_PLAN_COARSE_CYCLE, i=(1,), p=(self._engine._h.value,))

    def coarse_end(self, out_global):
        _lib.check(self.lib.mg_get_solution_device(self._engine._h, self._p(out_global), out_global.stride(0), self._code(out_global)))
        if self.rec is not None:
            self.rec.emit(_lib.MG_PLAN_COARSE_END, i=(out_global.stride(0), self._code(out_global)), p=(self._engine._h.value, out_global))

    def close(self):
        if self._engine is not None:
            self.torch.cuda.synchronize()
            self._engine.close()
            self._engine = None


# ------------------------------------------------------------------------------------------------
# the distributed driver
# ------------------------------------------------------------------------------------------------
class _Dom:
    """Per-rank state: one Block per distributed level and its fields."""


class _Phase:
    """`with solver._ph(name):` -- adds the time of the enclosed work to solver.phase_times[name] when profiling is on"""

    def __init__(self, owner, name):
        self.o, self.name = owner, name

    def __enter__(self):
        o = self.o
        self.on = o.phase_times is not None
        if not self.on:
            return self
        self.cuda = bool(getattr(o.ops, "supports_overlap", False))        # device kernels: stream-ordered timing events
        if self.cuda:
            self.e0 = o.torch.cuda.Event(enable_timing=True)
            self.e0.record()
        else:
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if not self.on:
            return False
        o = self.o
        if self.cuda:
            e1 = o.torch.cuda.Event(enable_timing=True)
            e1.record()
            o._phase_events.append((self.name, self.e0, e1))
        else:
            o.phase_times[self.name] += (time.perf_counter() - self.t0) * 1e3
        return False


# Ghost width of the fused mode: the smallest odd G for which the owned cells stay exact through every fused visit.
# With s = halo cells a leg's two sweeps consume (Jacobi 2, red-black GS 4: one per colour pass), m exact ghost cells
# after the up leg of a level and m_c on the level below: m = min(G - s, 2 m_c - 1) - s; the recursion must reproduce
# itself (m_c = m) and leave m >= 1 for the norm: Jacobi m = 3, G = 7; red-black GS m = 5, G = 13.
GHOST_FUSED = {"jacobi": 7, "rbgs": 13}


class DistributedMultigrid:
    """V/W/F-cycle on a px x py block decomposition.

    ranks:  the rank ids this PROCESS computes (one under torch.distributed; all of them for the
            in-process virtual-rank mode used by the single-GPU test).
    ops:    kernel provider (HipOps, or the tests' NumPy stand-in).
    dist:   torch.distributed module (initialised) or None for the in-process mode.
    mode:   "fused" (ghost zone of 7 cells for weighted Jacobi, 13 for red-black GS; two fused launches and ~one
            exchange per level; pre, post <= 2) or "per_operator" (1-cell ghost ring, one launch and one exchange per
            operator; any sweep count).  "auto" picks "fused" whenever it applies.
    """

    def __init__(self, NX, NY, px, py, ranks, ops, dist=None, domain=(0.0, 1.0, 0.0, 1.0), coeff=-1.0,
                 max_levels=None, cycle="V", pre=2, post=2, smoother="jacobi", omega=0.8, coarse_tol=1e-12,
                 coarse_maxit=1000, agglomerate_at=1025, mode="auto", overlap=True, native="auto", span="auto"):
        """span: run the level-0 up leg of cycle k and the down leg of cycle k + 1 as ONE launch (ops.span_leg) whenever the
        next cycle's front part is queued ahead of the norm anyway (`speculate`): "auto" with native plans (MG_DIST_SPAN=0
        turns it off), True also in the eager driver (tests), False never.  Weighted Jacobi, constant coefficients, level 0
        and level 1 in one dtype, blocks the kernel serves (ops.span_ok); same iterates bit for bit.
        native: replay the cycle from a recorded plan (dist_plan.py; one C call per cycle).  "auto": whenever the
        kernels are the device ones, the mode is "fused" and the ranks talk over RCCL (or live in this process).
        Precision: every level in ops.np_dtype, or -- with an `ops` built for per-level mixed precision (ops.mixed:
        PrecisionManager('mixed'), core/precision.py:337-357) -- level l >= L // 2 in fp32 and the rest, like the
        coarsest level, in fp64, decomposed and replicated levels alike."""
        from .facade import default_max_levels
        self.NX, self.NY, self.px, self.py = NX, NY, px, py
        self.ops, self.dist = ops, dist
        self.torch = ops.torch
        self.domain, self.coeff = domain, coeff
        self.cycle_type, self.pre, self.post = cycle, pre, post
        if smoother not in ("jacobi", "rbgs"):
            raise ValueError(f"Unknown smoother: {smoother}")
        if mode not in ("auto", "fused", "per_operator"):
            raise ValueError(f"Unknown mode: {mode}")
        can_fuse = pre <= 2 and post <= 2 and hasattr(ops, "down_leg")
        if mode == "fused" and not can_fuse:
            raise ValueError("mode 'fused' needs pre, post <= 2")
        self.mode = "fused" if (mode in ("auto", "fused") and can_fuse) else "per_operator"
        self.G = GHOST_FUSED[smoother] if self.mode == "fused" else 1
        self.smoother, self.omega = smoother, omega
        self.smk = _lib.MG_JACOBI if smoother == "jacobi" else _lib.MG_RBGS
        self.shapes = hierarchy_shapes(NX, NY, max_levels or default_max_levels(NX, NY))
        self.L = len(self.shapes)
        self.Ld = distributed_levels(self.shapes, px, py, agglomerate_at, self.G) if px * py > 1 else 0
        self.h = [((domain[1] - domain[0]) / (a - 1), (domain[3] - domain[2]) / (b - 1)) for a, b in self.shapes]
        # precision of every level (see the docstring); the coarsest level is never converted (solvers/multigrid.py:270-272)
        self.mixed = bool(getattr(ops, "mixed", False))
        self.split = self.L // 2
        self.ldt = [np.dtype(np.float32) if (self.mixed and l >= self.split and l != self.L - 1) else
                    (np.dtype(np.float64) if self.mixed else np.dtype(ops.np_dtype)) for l in range(self.L)]
        self.var = False                        # variable-coefficient operator (set_coefficient)
        self.ranks = list(ranks)
        self.doms = {}
        self.exchanges = 0                      # halo exchanges issued (statistics)
        for r in self.ranks:
            rx, ry = divmod(r, py)
            d = _Dom()
            d.rank, d.rx, d.ry = r, rx, ry
            d.blk = [Block(a, b, px, py, rx, ry, self.G) for (a, b) in self.shapes[:self.Ld + 1]]
            d.u, d.t, d.rhs, d.r, d.a, d.rd = [], [], [], [], [], []
            d.s = [None] * self.Ld                 # level 0 only: third buffer of the spanning leg, allocated on first use
            for l in range(self.Ld):
                b = d.blk[l]
                d.u.append(ops.alloc(b.lnx, b.lny, self.ldt[l]))
                d.t.append(ops.alloc(b.lnx, b.lny, self.ldt[l]) if (smoother == "jacobi" or self.mode == "fused") else None)
                d.rhs.append(ops.alloc(b.lnx, b.lny, self.ldt[l]))
                d.r.append(ops.alloc(b.lnx, b.lny, self.ldt[l]) if self.mode == "per_operator" else None)
                d.a.append(None)
                d.rd.append(None)
            # the agglomeration level: a local coarse buffer (restriction target / prolongation source)
            if self.Ld > 0:
                b = d.blk[self.Ld]
                d.rc = ops.alloc(b.lnx, b.lny, self.ldt[self.Ld])   # restricted rhs (its boundary ring is written once per rhs in fused mode)
                d.ec = ops.alloc(b.lnx, b.lny, self.ldt[self.Ld])   # this rank's piece of the replicated correction
                d.zc = None                                          # zero correction for the level-1 block (variable-coefficient norm)
            d.ring_sumsq = None
            self.doms[r] = d
        NXa, NYa = self.shapes[self.Ld]
        self.rhs_a = ops.alloc(NXa, NYa, self.ldt[self.Ld])
        self.e_a = ops.alloc(NXa, NYa, self.ldt[self.Ld])
        ops.coarse_setup(NXa, NYa, domain, dict(coeff=coeff, levels=self.L - self.Ld, cycle=cycle, pre=pre, post=post,
                                                smoother=self.smk, omega=omega, coarse_tol=coarse_tol, coarse_maxit=coarse_maxit,
                                                mixed_split=self.split - self.Ld))
        # gather buffers: exclusive blocks padded to the largest block
        if self.Ld > 0:
            blocks = [Block(NXa, NYa, px, py, rx, ry, self.G) for rx in range(px) for ry in range(py)]
            self.gmx = max(b.i_hi - b.i_lo for b in blocks)
            self.gmy = max(b.j_hi - b.j_lo for b in blocks)
        self._last_norm_parts = None
        self._stage_p2p = None                   # decided at the first exchange (see _p2p)
        # exchange / compute overlap on a second stream (device kernels only)
        self.overlap = bool(overlap) and self.mode == "fused" and getattr(ops, "supports_overlap", False)
        if self.overlap:
            torch = self.torch
            from . import dist_plan
            # ONE communication stream per process and device: the solvers of a process share the RCCL communicator, and every
            # RCCL call on a communicator must reach it from one stream (dist_plan.shared_comm_stream)
            self._comm_stream = dist_plan.shared_comm_stream(getattr(ops, "device", torch.device("cuda", torch.cuda.current_device())))
            self._ev_a, self._ev_b = torch.cuda.Event(), torch.cuda.Event()
            self._ev_c, self._ev_d = torch.cuda.Event(), torch.cuda.Event()      # spanning mode: the level-0 exchange beside the lower levels
        # native replay of the cycle (dist_plan.py)
        plan_ok = (self.mode == "fused" and getattr(ops, "plan_capable", False) and self.Ld > 0 and
                   (dist is None or dist.get_backend() == "nccl"))
        if native not in ("auto", True, False):
            raise ValueError(f"Unknown native setting: {native}")
        if native is True and not plan_ok:
            raise ValueError("native cycle plans need the device kernels, mode 'fused' and RCCL (or in-process ranks)")
        self.native = plan_ok if native == "auto" else bool(native)
        self.native_required = native is True
        if span not in ("auto", True, False):
            raise ValueError(f"Unknown span setting: {span}")
        self.span = (self.native and os.environ.get("MG_DIST_SPAN", "1") != "0") if span == "auto" else bool(span)
        self._pre = None                         # spanning mode: the level-0 buffer ("t" / "s") holding the queued front part's pre-smoothed iterate
        self._sp_plans = {}                      # ... its recorded plans: ("mid", src) -> (legs + norm, lower levels), ("back", src) -> plan
        self._norm_plan = None                   # the plan whose RESULT (sum of r^2) is in flight
        self._plan_x = {}                        # plan -> halo exchanges it issues (statistics)
        self._plan_kind = None                   # which scheme self._plan belongs to: "plain" (front | back) or "span"
        self._plan_failure = None                # why the last recorded part has no plan (spanning scheme)
        self.native_failure = None               # why "auto" fell back to the Python driver, if it did
        self._plan_exchanges = 0
        self._rec = None                         # PlanRecorder while the first cycle is being recorded
        self._plan = None
        self._plan_state = None
        self._comm = None
        self._bufs = {}                          # persistent staging buffers (pack / unpack / gather)
        self._norm_value = None                  # sum of r^2 the last native cycle returned
        self._plan_back = None                   # the plan is kept as two: front (level-0 down legs and below) / back (up legs, norm)
        self._front_queued = False               # the front part of the COMING cycle is already on the streams
        self._norm_pending = False               # the back part's sum of r^2 is still on its way to the host
        self.speculate = True                    # queue the next cycle's front part before waiting for the norm
        self.native_cycles = 0
        # per-phase times of the cycle (diagnostics, profile_phases / collect_phase_times)
        self.phase_times = None
        self._phase_events = []

    # ---- per-phase times (diagnostics) -----------------------------------------------------------
    def profile_phases(self, enable=True):
        """Bracket the phases of the coming cycles -- fused legs, halo copies, send/recv groups (incl. the wait for the
        peers), the coarse all-gather, the replicated engine, the norm all-reduce -- with timers: timing events on the stream a
        phase runs on for device tensors (inside the C++ plan executor when cycles are replayed natively), wall clock for the
        CPU stand-in.  For diagnostic cycles outside a timed region; collect_phase_times() returns milliseconds per phase."""
        self.phase_times = {n: 0.0 for n in _lib.PLAN_PHASE_NAMES} if enable else None
        self._phase_events = []
        for pl in self._all_plans():
            pl.profile(enable)

    def collect_phase_times(self):
        if self.phase_times is None:
            return {}
        if self._phase_events:
            self.torch.cuda.synchronize()
            for name, e0, e1 in self._phase_events:
                self.phase_times[name] += e0.elapsed_time(e1)
            self._phase_events = []
        for pl in self._all_plans():
            pl.phase_times(self.phase_times)
        out = dict(self.phase_times)
        for k in self.phase_times:
            self.phase_times[k] = 0.0
        return out

    def _ph(self, name):
        return _Phase(self, name)

    # ---- primitives the plan recorder sees ------------------------------------------------------
    def _copy(self, dst, src):
        dst.copy_(src)
        if self._rec is not None:
            self._rec.copy2d(dst, src)

    def _buf(self, key, shape, dtype, device, zero=False):
        """staging buffer that lives as long as the solver (a recorded plan holds its pointer)"""
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (self.torch.zeros if zero else self.torch.empty)(tuple(shape), dtype=dtype, device=device)
            self._bufs[key] = t
        return t

    def _add(self, a, b):
        out = a + b
        if self._rec is not None:
            self._rec.add(out, a, b)
        return out

    # ---- neighbours ----------------------------------------------------------------------
    def _nbr(self, d, dx, dy):
        rx, ry = d.rx + dx, d.ry + dy
        if 0 <= rx < self.px and 0 <= ry < self.py:
            return rx * self.py + ry
        return None

    def _p2p(self, sends, recvs):
        """sends/recvs: lists of (peer_rank, tensor): batched isend/irecv (RCCL send/recv, one group per phase)."""
        if self.dist is None or not (sends or recvs):
            return
        if self._rec is not None:
            self._rec.group(sends, recvs)
        if self._stage_p2p is None:
            # gloo moves device tensors with host-side memcpy on their raw pointers, unordered against the HIP streams
            # that produce / consume them (rehearsals on a one-GPU box): stage through host tensors, synchronously.
            # nccl (= RCCL) send/recv are enqueued on the current stream and need nothing of the kind.
            self._stage_p2p = (self.dist.get_backend() == "gloo") and any(t.is_cuda for _, t in sends + recvs)
        if self._stage_p2p:
            self.torch.cuda.current_stream().synchronize()
            hs = [(p, t.cpu()) for p, t in sends]
            hr = [(p, self.torch.empty(t.shape, dtype=t.dtype), t) for p, t in recvs]
            ops = [self.dist.P2POp(self.dist.isend, t, p) for p, t in hs] + \
                  [self.dist.P2POp(self.dist.irecv, h, p) for p, h, _ in hr]
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
            for _, h, t in hr:
                t.copy_(h)
            return
        ops = [self.dist.P2POp(self.dist.isend, t, p) for p, t in sends] + \
              [self.dist.P2POp(self.dist.irecv, t, p) for p, t in recvs]
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()

    def exchange(self, name, l, corners=False):
        """Fill the ghost zone (G cells wide) of field `name` on level l from the neighbours' owned cells next to the cut,
        in ONE communication step: whole rows to the row neighbours (contiguous memory, no packing), packed columns over
        the full local height to the column neighbours, packed G x G corners to the diagonal neighbours.  Rows and columns
        carry stale data where they cross the receiver's corner regions; the corner blocks are unpacked last."""
        G = self.G
        self.exchanges += 1
        fields = {r: (getattr(d, name)[l] if isinstance(getattr(d, name), list) else getattr(d, name)) for r, d in self.doms.items()}

        def rows_of(b, dx, ghost):       # the G rows next to the cut towards dx: the ghost rows, or the owned rows beside them
            if ghost:
                return slice(b.oi_lo - G, b.oi_lo) if dx < 0 else slice(b.oi_hi + 1, b.oi_hi + 1 + G)
            return slice(b.oi_lo, b.oi_lo + G) if dx < 0 else slice(b.oi_hi - G + 1, b.oi_hi + 1)

        def cols_of(b, dy, ghost):
            if ghost:
                return slice(b.oj_lo - G, b.oj_lo) if dy < 0 else slice(b.oj_hi + 1, b.oj_hi + 1 + G)
            return slice(b.oj_lo, b.oj_lo + G) if dy < 0 else slice(b.oj_hi - G + 1, b.oj_hi + 1)

        sends, recvs = [], []
        # in-process neighbours: copies, grouped so that runs of them do not touch each other (the plan executor launches such
        # a run as one kernel): all rows, all columns, then the corners (they overwrite what rows and columns left there)
        local = {"row": [], "col": [], "corner": []}
        unpack = {"col": [], "corner": []}
        for r, d in self.doms.items():
            b, t = d.blk[l], fields[r]
            for dx in (-1, 0, +1):
                for dy in (-1, 0, +1):
                    if dx == 0 and dy == 0:
                        continue
                    p = self._nbr(d, dx, dy)
                    if p is None:
                        continue
                    pb = self.doms[p].blk[l] if p in self.doms else None
                    if dy == 0:                                   # row neighbour
                        if pb is not None:
                            local["row"].append((t[rows_of(b, dx, True), :b.lny], fields[p][rows_of(pb, -dx, False), :pb.lny]))
                        else:
                            sends.append((p, t[rows_of(b, dx, False)]))      # whole padded rows: one contiguous chunk
                            recvs.append((p, t[rows_of(b, dx, True)]))
                    elif dx == 0:                                 # column neighbour
                        if pb is not None:
                            local["col"].append((t[:b.lnx, cols_of(b, dy, True)], fields[p][:pb.lnx, cols_of(pb, -dy, False)]))
                        else:
                            sbuf = self._buf(("pack", name, l, r, dy), (b.lnx, G), t.dtype, t.device)
                            rbuf = self._buf(("unpack", name, l, r, dy), (b.lnx, G), t.dtype, t.device)
                            self._copy(sbuf, t[:b.lnx, cols_of(b, dy, False)])
                            sends.append((p, sbuf))
                            recvs.append((p, rbuf))
                            unpack["col"].append((t[:b.lnx, cols_of(b, dy, True)], rbuf))
                    else:                                         # diagonal neighbour: the G x G corner
                        dst = t[rows_of(b, dx, True), cols_of(b, dy, True)]
                        if pb is not None:
                            local["corner"].append((dst, fields[p][rows_of(pb, -dx, False), cols_of(pb, -dy, False)]))
                        else:
                            sbuf = self._buf(("cpack", name, l, r, dx, dy), (G, G), t.dtype, t.device)
                            rbuf = self._buf(("cunpack", name, l, r, dx, dy), (G, G), t.dtype, t.device)
                            self._copy(sbuf, t[rows_of(b, dx, False), cols_of(b, dy, False)])
                            sends.append((p, sbuf))
                            recvs.append((p, rbuf))
                            unpack["corner"].append((dst, rbuf))
        with self._ph("halo_copy"):
            for dst, src in local["row"] + local["col"]:
                self._copy(dst, src)
        with self._ph("halo_exchange"):
            self._p2p(sends, recvs)
        with self._ph("halo_copy"):
            for dst, src in unpack["col"] + local["corner"] + unpack["corner"]:
                self._copy(dst, src)

    def allreduce_sum(self, parts):
        """parts: {rank: 1-element fp64 tensor}.  Returns the global sum as a Python float."""
        total = None
        with self._ph("allreduce"):
            for r in self.ranks:
                total = parts[r] if total is None else self._add(total, parts[r])
            if self.dist is not None:
                self.dist.all_reduce(total)
                if self._rec is not None:
                    self._rec.allreduce(total)
        if self._rec is not None:
            self._rec.result(total)
        return float(total.item())

    # ---- agglomeration -----------------------------------------------------------------------
    def _gather_coarse_rhs(self):
        """All ranks end up with the whole coarse right-hand side (exclusive blocks tile the grid)."""
        torch = self.torch
        La = self.Ld
        NXa, NYa = self.shapes[La]
        if self.dist is None:
            for r, d in self.doms.items():
                b = d.blk[La]
                self._copy(self.rhs_a[b.gx0 + b.i_lo:b.gx0 + b.i_hi, b.gy0 + b.j_lo:b.gy0 + b.j_hi], d.rc[b.i_lo:b.i_hi, b.j_lo:b.j_hi])
            return
        (r, d), = self.doms.items()
        b = d.blk[La]
        P = self.px * self.py
        mine = self._buf(("gather", "mine"), (self.gmx, self.gmy), d.rc.dtype, d.rc.device, zero=True)   # the padding stays zero
        every = self._buf(("gather", "all"), (P, self.gmx, self.gmy), d.rc.dtype, d.rc.device)
        self._copy(mine[:b.i_hi - b.i_lo, :b.j_hi - b.j_lo], d.rc[b.i_lo:b.i_hi, b.j_lo:b.j_hi])
        self.dist.all_gather(list(every.unbind(0)), mine)
        if self._rec is not None:
            self._rec.allgather(mine, every)
        for q in range(P):
            qb = Block(NXa, NYa, self.px, self.py, *divmod(q, self.py), self.G)
            self._copy(self.rhs_a[qb.gx0 + qb.i_lo:qb.gx0 + qb.i_hi, qb.gy0 + qb.j_lo:qb.gy0 + qb.j_hi],
                       every[q, :qb.i_hi - qb.i_lo, :qb.j_hi - qb.j_lo])

    def _replicated_cycle(self, l):
        """Coarse tail: gather the coarse rhs, run the remaining levels on the single-GPU engine (on every GPU),
        take this rank's piece of the correction (ghost zone included: it is global data)."""
        with self._ph("coarse_allgather"):
            self._gather_coarse_rhs()
        with self._ph("replicated_engine"):
            if self.mode == "fused" and getattr(self.ops, "plan_capable", False):
                self.ops.coarse_begin(self.rhs_a, same_ring=True)      # the ring went in with set_problem
            else:
                self.ops.coarse_begin(self.rhs_a)
            for _ in range(self._reps(l)):
                self.ops.coarse_cycle()
            self.ops.coarse_end(self.e_a)
        with self._ph("halo_copy"):
            for d in self.doms.values():
                bc = d.blk[l + 1]
                self._copy(d.ec[:bc.lnx, :bc.lny], self.e_a[bc.gx0:bc.gx0 + bc.lnx, bc.gy0:bc.gy0 + bc.lny])

    # ---- the cycle (solvers/multigrid.py:253-337) ------------------------------------------------
    def _reps(self, l):
        if self.cycle_type == "V":
            return 1
        if self.cycle_type == "W":
            return 2
        return max(1, 2 ** (self.L - l - 2))

    def smooth(self, l, nu):
        hx, hy = self.h[l]
        for _ in range(nu):
            if self.smoother == "jacobi":
                for d in self.doms.values():
                    b = d.blk[l]
                    self.ops.jacobi(d.u[l], d.rhs[l], d.t[l], b.lnx, b.lny, hx, hy, self.omega)
                    d.u[l], d.t[l] = d.t[l], d.u[l]
                self.exchange("u", l)
            else:
                for colour in (0, 1):
                    for d in self.doms.values():
                        b = d.blk[l]
                        self.ops.rbgs_colour(d.u[l], d.rhs[l], b.lnx, b.lny, hx, hy, self.omega, colour,
                                             (b.gx0 + b.gy0) & 1)
                    self.exchange("u", l)

    def cycle(self, l=0, zero_u=False):
        if self.Ld == 0:                      # nothing distributed: the replicated engine is the whole solver
            raise RuntimeError("single-block problems go through MultigridEngine")
        self._last_norm_parts = None
        self._norm_value = None
        spanning = l == 0 and not zero_u and self._span_usable()
        if self.native and l == 0 and not zero_u:
            return self._cycle_native_span() if spanning else self._cycle_native()
        if spanning:
            if self._norm_pending or self._front_queued:      # native cycles came before: start from their iterate
                self._settle()
            return self._cycle_span_eager()
        # an eager cycle after native ones: collect a norm still in flight and forget a queued front part -- it was computed
        # from the iterate this cycle is about to replace
        self._settle()
        if self.mode == "fused":
            return self._cycle_fused(l, zero_u)
        return self._cycle_per_operator(l)

    # ---- native replay (dist_plan.py) --------------------------------------------------------------
    def _pointer_state(self):
        return tuple((d.u[l].data_ptr(), d.t[l].data_ptr(), d.rhs[l].data_ptr(), 0 if d.a[l] is None else d.a[l].data_ptr(),
                      0 if d.rd[l] is None else d.rd[l].data_ptr(), 0 if d.s[l] is None else d.s[l].data_ptr())
                     for d in self.doms.values() for l in range(self.Ld)) + \
            tuple(0 if d.ring_sumsq is None else d.ring_sumsq.data_ptr() for d in self.doms.values()) + (self.var,)

    def _all_plans(self):
        out = [q for q in (self._plan, self._plan_back) if q is not None]
        for pair in self._sp_plans.values():
            out.extend(q for q in pair if q is not None)
        return out

    def _drop_plan(self):
        plans = self._all_plans()
        if plans:
            self.torch.cuda.synchronize()
            for q in plans:
                q.close()
        self._plan = self._plan_back = self._norm_plan = None
        self._sp_plans = {}
        self._plan_x = {}
        self._plan_state = None
        self._front_queued = self._norm_pending = False
        self._pre = None

    def _settle(self):
        """Before the iterate is replaced (new problem / coefficient / iterate): collect a norm still in flight and forget a
        front part queued for a cycle that will not come."""
        if self._norm_pending:
            self._norm_value = (self._norm_plan or self._plan_back).wait()
            self._norm_pending = False
        self._front_queued = False
        self._pre = None                         # spanning mode: u holds the iterate, whatever t / s were being prepared for

    def _record_part(self, fn):
        """run `fn` through the Python driver with a recorder attached -> CyclePlan (None, and self._plan_failure, if it cannot be built)"""
        from . import dist_plan
        device = next(iter(self.doms.values())).u[0].device
        before = self.exchanges
        self._rec = self.ops.rec = dist_plan.PlanRecorder()
        try:
            fn()
            rec = self._rec
        finally:
            self._rec = self.ops.rec = None
        try:
            if self.dist is not None and self._comm is None:
                self._comm = dist_plan.shared_comm(self.dist, device.index or 0)
            plan = dist_plan.CyclePlan(rec, self._comm, device.index or 0)
        except Exception as exc:                 # the work is done; only the replay is missing (agreed on by all ranks below)
            self._plan_failure = exc
            return None
        self._plan_x[plan] = self.exchanges - before
        if self.phase_times is not None:
            plan.profile(True)
        return plan

    def _cycle_native_span(self):
        """The spanning scheme with recorded plans: front F (u -> t, lower levels), and per source buffer a mid pair (spanning
        legs + norm | exchange + lower levels) and a back plan (up legs + norm).  Each is recorded from the Python driver the
        first time its turn comes (that cycle runs eagerly) and replayed afterwards; the norm travels behind the legs' plan."""
        torch = self.torch
        if self._plan is not None and self._plan_kind != "span":      # plans of the two-launch scheme
            self._settle()
            self._drop_plan()
        self._ensure_third()
        state = self._pointer_state()
        if (self._plan is not None or self._sp_plans) and self._plan_state != state:
            self._settle()
            self._drop_plan()
        comm = self._comm_stream.cuda_stream if self.overlap else torch.cuda.current_stream().cuda_stream
        compute = torch.cuda.current_stream().cuda_stream
        if self._norm_pending:                   # nobody asked for the previous cycle's norm
            (self._norm_plan or self._plan_back).wait()
            self._norm_pending = False
        replayed = True
        self._plan_failure = None
        if self._pre is None:
            if self._plan is None:
                self._plan = self._record_part(lambda: (self._sp_front("t"), self._sp_lower("t")))
                self._plan_state, self._plan_kind = state, "span"
                replayed = False
            else:
                self._plan.run_async(compute, comm)
                self.exchanges += self._plan_x[self._plan]
            self._pre = "t"
        src = self._pre
        dst = ("s" if src == "t" else "t") if self.speculate else None
        key = ("mid" if dst else "back", src)
        plans = self._sp_plans.get(key)
        if plans is None:
            def legs():
                self._sp_legs(src, dst)
                self._norm_value = self.allreduce_sum(self._last_norm_parts)
            first = self._record_part(legs)
            self._last_norm_parts = None
            second = self._record_part(lambda: self._sp_lower(dst)) if dst else None
            self._sp_plans[key] = (first, second)
            replayed = False
        else:
            for q in plans:
                if q is not None:
                    q.run_async(compute, comm)
                    self.exchanges += self._plan_x[q]
            self._norm_plan, self._norm_pending = plans[0], True
        self._pre = dst
        if not replayed:
            # a rank that could not build a plan takes every rank back to the Python driver -- agreed on through
            # torch.distributed (every rank records in the same cycle), so nobody replays alone
            failure = self._plan_failure
            if self.dist is not None:
                device = next(iter(self.doms.values())).u[0].device
                flag = torch.tensor([0 if failure is None else 1], dtype=torch.int32, device=device)
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
                if int(flag.item()) and failure is None:
                    failure = RuntimeError("another rank could not build its cycle plan")
            if failure is not None:
                if self.native_required:
                    raise failure
                norm = self._norm_value
                self._drop_plan()                # also forgets the queued front part: the eager driver starts from the iterate in u
                self._norm_value = norm
                self.native = False
                self.native_failure = repr(failure)
                return
        self._front_queued = self._pre is not None
        if replayed:
            self.native_cycles += 1

    def _cycle_native(self):
        """The first cycle (and the first after anything moved a field to another buffer) runs through the Python driver
        with a recorder attached; every other one is a single mg_plan_run.  Either way the sum of r^2 over the grid comes
        back with the cycle (the eager path computes it lazily in residual_norm())."""
        from . import dist_plan
        torch = self.torch
        state = self._pointer_state()
        if self._plan is not None and (self._plan_state != state or self._plan_kind == "span"):
            self._settle()
            self._drop_plan()
        self._plan_kind = "plain"
        if self._plan is None:
            device = next(iter(self.doms.values())).u[0].device
            before = self.exchanges
            self._rec = self.ops.rec = dist_plan.PlanRecorder()
            try:
                self._cycle_fused(0, False)
                self._norm_value = self.allreduce_sum(self._last_norm_parts)
                rec = self._rec
            finally:
                self._rec = self.ops.rec = None
            self._last_norm_parts = None
            self._plan_exchanges = self.exchanges - before
            if self._pointer_state() != state:
                raise RuntimeError("a cycle must leave every field in the buffer it started in")
            # the plan (and, between processes, the library's own RCCL communicator); a rank that cannot build one takes
            # every rank back to the Python driver -- agreed on through torch.distributed, so nobody replays alone
            failure = None
            try:
                if self.dist is not None and self._comm is None:
                    self._comm = dist_plan.shared_comm(self.dist, device.index or 0)
                split = rec.split if rec.split is not None else len(rec.ops)
                self._plan = dist_plan.CyclePlan(rec, self._comm, device.index or 0, 0, split)          # front part
                self._plan_back = dist_plan.CyclePlan(rec, self._comm, device.index or 0, split, None)
                self._plan_state = state
                if self.phase_times is not None:
                    self._plan.profile(True)
                    self._plan_back.profile(True)
            except Exception as exc:
                failure = exc
            if self.dist is not None:
                flag = torch.tensor([0 if failure is None else 1], dtype=torch.int32, device=device)
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
                if int(flag.item()) and failure is None:
                    failure = RuntimeError("another rank could not build its cycle plan")
            if failure is not None:
                if self.native_required:
                    raise failure
                self._drop_plan()
                self.native = False
                self.native_failure = repr(failure)
            return
        comm = self._comm_stream.cuda_stream if self.overlap else torch.cuda.current_stream().cuda_stream
        compute = torch.cuda.current_stream().cuda_stream
        if self._norm_pending:                   # nobody asked for the previous cycle's norm
            self._plan_back.wait()
            self._norm_pending = False
        if not self._front_queued:
            self._plan.run_async(compute, comm)
        self._plan_back.run_async(compute, comm)
        self._norm_pending = True
        # the front part of the NEXT cycle goes onto the streams before anybody waits for this cycle's norm: it reads the
        # iterate this cycle leaves and writes only the other buffers, so a solve that ends here just leaves it unused
        self._front_queued = bool(self.speculate)
        if self._front_queued:
            self._plan.run_async(compute, comm)
        self.native_cycles += 1
        self.exchanges += self._plan_exchanges

    def _down_legs(self, l, zero_u, pending, out=None):
        """The down legs of level l on every local block: u -> `out` (rank -> array; default the ping-pong partner t), the
        restricted residual -> the level below.  `pending`: fields of this level whose ghost zones the legs wait for; with
        overlap the exchange runs on the communication stream beside the tiles that read no ghost data."""
        hx, hy = self.h[l]
        last = (l + 1 == self.Ld)

        def down(select, inner_of):
            for r, d in self.doms.items():
                b, bc = d.blk[l], d.blk[l + 1]
                ci, cj = b.coarse_offsets(bc)
                target = d.rc if last else d.rhs[l + 1]
                kw = {"acoef": d.a[l], "rdiag": d.rd[l]} if self.var else {}
                with self._ph("legs"):
                    self.ops.down_leg(self.smk, d.u[l], d.rhs[l], d.t[l] if out is None else out[r], target, b.lnx, b.lny, bc.lnx, bc.lny,
                                      ci, cj, hx, hy, self.omega, self.coeff, self.pre, zero_u, (b.gx0 + b.gy0) & 1,
                                      select, inner_of(b) if select else None, **kw)

        def inner_rect(b):      # cells whose values do not come out of an exchange: owned cells and physical boundary
            big = 1 << 30
            return (-big if b.sides & SIDE_ILO else b.oi_lo, big if b.sides & SIDE_IHI else b.oi_hi + 1,
                    -big if b.sides & SIDE_JLO else b.oj_lo, big if b.sides & SIDE_JHI else b.oj_hi + 1)

        if pending and self.overlap:
            # tiles that read no ghost data run on the compute stream while the exchange runs on the comm stream
            torch = self.torch
            compute = torch.cuda.current_stream()
            rec = self._rec
            self._ev_a.record(compute)
            if rec is not None:
                rec.event_record(0)
                rec.stream = 1
                rec.stream_wait(0)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(self._ev_a)
                for name in pending:
                    self.exchange(name, l)
                self._ev_b.record(self._comm_stream)
            if rec is not None:
                rec.event_record(1)
                rec.stream = 0
            down(1, inner_rect)
            compute.wait_event(self._ev_b)
            if rec is not None:
                rec.stream_wait(1)
            down(2, inner_rect)
        else:
            for name in pending:
                self.exchange(name, l)
            down(0, None)

    # ---- level 0 with a spanning leg ------------------------------------------------------------------------------------
    # Three level-0 arrays per block: u ALWAYS holds the iterate of the last completed cycle; t and s take turns holding the
    # pre-smoothed iterate of the cycle in flight (self._pre names the one that does).  front: u -> t (down legs);
    # mid: src -> u (iterate of this cycle) and -> the other one (pre-smoothed iterate of the next), one launch; back: src -> u
    # (up legs).  Every part ends with the level-0 ghost zones of what it just wrote on their way (beside the lower levels), so
    # that the part that follows -- mid or back -- reads m = 7 exact ghost cells: post sweeps leave 4, the norm reads owned
    # cells, pre sweeps leave 2, residual 1, full weighting of the owned coarse cells needs 1.
    def _f0(self, d, name):
        return d.u[0] if name == "u" else (d.t[0] if name == "t" else d.s[0])

    def _span_usable(self):
        if not (self.span and self.mode == "fused" and self.smoother == "jacobi" and not self.var and self.Ld >= 1 and
                1 <= self.pre <= 2 and 1 <= self.post <= 2 and hasattr(self.ops, "span_leg") and self.ldt[0] == self.ldt[1]):
            return False
        for d in self.doms.values():
            b = d.blk[0]
            if not self.ops.span_ok(self.smk, d.u[0], d.ec if self.Ld == 1 else d.u[1], b.lnx, b.lny):
                return False
        return True

    def _ensure_third(self):
        for d in self.doms.values():
            if d.s[0] is None:
                b = d.blk[0]
                d.s[0] = self.ops.alloc(b.lnx, b.lny, self.ldt[0])
                d.s[0].copy_(d.u[0])               # the outermost ring (Dirichlet values on physical edges) is never written by a leg

    def _sp_front(self, dst):
        self._down_legs(0, False, ["u"], out={r: self._f0(d, dst) for r, d in self.doms.items()})

    def _sp_lower(self, xname):
        """everything below level 0 of one cycle, the halo exchange of the level-0 array `xname` beside it"""
        torch, rec = self.torch, self._rec
        if self.overlap:
            compute = torch.cuda.current_stream()
            self._ev_c.record(compute)
            if rec is not None:
                rec.event_record(2)
                rec.stream = 1
                rec.stream_wait(2)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(self._ev_c)
                self.exchange(xname, 0)
                self._ev_d.record(self._comm_stream)
            if rec is not None:
                rec.event_record(3)
                rec.stream = 0
        else:
            self.exchange(xname, 0)
        if self.Ld == 1:
            self._replicated_cycle(0)
        else:
            for k in range(self._reps(0)):
                self._cycle_fused(1, k == 0, k == 0)
        if self.overlap:
            compute.wait_event(self._ev_d)
            if rec is not None:
                rec.stream_wait(3)

    def _sp_legs(self, src, dst):
        """dst None: the up legs src -> u (back part); else the spanning legs src -> u and dst (mid part).  Sets the norm parts."""
        hx, hy = self.h[0]
        parts = {}
        for r, d in self.doms.items():
            b, bc = d.blk[0], d.blk[1]
            ci, cj = b.coarse_offsets(bc)
            last = (self.Ld == 1)
            e = d.ec if last else d.u[1]
            win = (max(b.i_lo, 1), min(b.i_hi, b.lnx - 1), max(b.j_lo, 1), min(b.j_hi, b.lny - 1))
            with self._ph("legs"):
                if dst is None:
                    res = self.ops.up_leg(self.smk, self._f0(d, src), d.rhs[0], d.u[0], e, b.lnx, b.lny, bc.lnx, bc.lny, ci, cj, b.sides,
                                          hx, hy, self.omega, self.coeff, self.post, (b.gx0 + b.gy0) & 1, win)
                else:
                    res = self.ops.span_leg(self.smk, self._f0(d, src), d.rhs[0], d.u[0], self._f0(d, dst), e, d.rc if last else d.rhs[1],
                                            b.lnx, b.lny, bc.lnx, bc.lny, ci, cj, b.sides, hx, hy, self.omega, self.coeff, self.post,
                                            self.pre, (b.gx0 + b.gy0) & 1, win)
            parts[r] = self._add(res, d.ring_sumsq)
        self._last_norm_parts = parts

    def _cycle_span_eager(self):
        """one cycle of the spanning scheme through the Python driver (the order of operations the plans replay)"""
        self._ensure_third()
        if self._pre is None:
            self._sp_front("t")
            self._sp_lower("t")
            self._pre = "t"
        if self.speculate:                           # the next cycle's front part goes out with this cycle's back part
            dst = "s" if self._pre == "t" else "t"
            self._sp_legs(self._pre, dst)
            parts = self._last_norm_parts
            self._sp_lower(dst)
            self._last_norm_parts = parts
            self._pre = dst
        else:
            self._sp_legs(self._pre, None)
            self._pre = None

    def _cycle_fused(self, l, zero_u, first_visit_rhs=False):
        """Two launches and (at most) two exchanges per level.  Validity bookkeeping (m = cells of the ghost zone that
        are exact, counted from the owned cells outwards; G = 7): after an exchange m = 7; the down leg's two sweeps
        leave the iterate exact on m = 5, its restriction is exact on all owned coarse cells; the correction that
        comes back from below is exact on m_c >= 3 coarse cells = 6 fine cells, so after the up leg (prolongation,
        two sweeps) m = min(5, 6) - 2 = 3 >= 0, and the norm (one more cell) only reads exact values."""
        hx, hy = self.h[l]
        last = (l + 1 == self.Ld)
        # What this level's down leg is waiting for: the iterate's ghost zone (level 0 every cycle; coarser levels only
        # when re-visited by a W / F cycle) and, below level 0, the ghost zone of the rhs the level above just produced.
        pending = []
        if not zero_u:
            pending.append("u")
        if l > 0 and first_visit_rhs:
            pending.append("rhs")

        self._down_legs(l, zero_u, pending)
        for d in self.doms.values():
            d.u[l], d.t[l] = d.t[l], d.u[l]
        if last:
            self._replicated_cycle(l)
        else:
            for k in range(self._reps(l)):
                self._cycle_fused(l + 1, k == 0, k == 0)
        want_norm = (l == 0)
        if l == 0 and self._rec is not None:
            self._rec.mark_split()               # what follows -- the level-0 up legs and the norm -- is the back part
        parts = {}
        for r, d in self.doms.items():
            b, bc = d.blk[l], d.blk[l + 1]
            ci, cj = b.coarse_offsets(bc)
            e = d.ec if last else d.u[l + 1]
            win = (max(b.i_lo, 1), min(b.i_hi, b.lnx - 1), max(b.j_lo, 1), min(b.j_hi, b.lny - 1)) if want_norm else None
            kw = {"acoef": d.a[l], "rdiag": d.rd[l]} if self.var else {}
            with self._ph("legs"):
                res = self.ops.up_leg(self.smk, d.u[l], d.rhs[l], d.t[l], e, b.lnx, b.lny, bc.lnx, bc.lny, ci, cj, b.sides, hx, hy,
                                      self.omega, self.coeff, self.post, (b.gx0 + b.gy0) & 1, win, **kw)
            d.u[l], d.t[l] = d.t[l], d.u[l]
            if want_norm:
                parts[r] = self._add(res, d.ring_sumsq)
        if want_norm:
            self._last_norm_parts = parts

    def _cycle_per_operator(self, l):
        if self.var:
            raise NotImplementedError("the variable-coefficient operator runs on the fused legs (pre, post <= 2)")
        hx, hy = self.h[l]
        if self.pre > 0:
            self.smooth(l, self.pre)
        for d in self.doms.values():
            b = d.blk[l]
            self.ops.residual(d.u[l], d.rhs[l], d.r[l], b.lnx, b.lny, hx, hy, self.coeff)
        self.exchange("r", l, corners=True)
        last = (l + 1 == self.Ld)
        for d in self.doms.values():
            b, bc = d.blk[l], d.blk[l + 1]
            target = d.rc if last else d.rhs[l + 1]
            self.ops.restrict(d.r[l], target, b.lnx, b.lny, bc.lnx, bc.lny, bc.sides)
        if last:
            self._replicated_cycle(l)
            for d in self.doms.values():
                b, bc = d.blk[l], d.blk[l + 1]
                self.ops.prolong_add(d.ec, d.u[l], b.lnx, b.lny, bc.lnx, bc.lny, b.sides)
        else:
            for d in self.doms.values():
                d.u[l + 1].zero_()
            for _ in range(self._reps(l)):
                self._cycle_per_operator(l + 1)
            for d in self.doms.values():
                b, bc = d.blk[l], d.blk[l + 1]
                self.ops.prolong_add(d.u[l + 1], d.u[l], b.lnx, b.lny, bc.lnx, bc.lny, b.sides)
        if self.post > 0:
            self.smooth(l, self.post)

    # ---- fields in / out -------------------------------------------------------------------------
    def set_coefficient(self, a_at):
        """Variable-coefficient operator A = coeff * div(a grad .) (BASELINE config 5; not in the reference).
        a_at(ix, iy) -> 2-D array of the vertex values of a at GLOBAL FINE-grid indices (1-D integer arrays ix, iy).
        Level l takes every 2^l-th fine vertex (injection, the single-GPU engine's rule), so every rank fills its blocks
        of every level -- ghost zones included -- from the function alone: no exchange.  None: constant coefficients."""
        torch = self.torch
        self._settle()
        if a_at is None:
            self.var = False
            self.ops.coarse_coefficient(None)
            return
        if self.mode != "fused":
            raise NotImplementedError("the variable-coefficient operator runs on the fused legs (pre, post <= 2)")
        for d in self.doms.values():
            for l in range(self.Ld):
                b = d.blk[l]
                vals = np.asarray(a_at((b.gx0 + np.arange(b.lnx)) << l, (b.gy0 + np.arange(b.lny)) << l))
                if d.a[l] is None:
                    d.a[l] = self.ops.alloc(b.lnx, b.lny, self.ldt[l])
                    d.rd[l] = self.ops.alloc(b.lnx, b.lny, self.ldt[l])
                d.a[l][:b.lnx, :b.lny] = torch.as_tensor(np.ascontiguousarray(vals, dtype=self.ldt[l])).to(d.a[l].device)
                # the reciprocal diagonal of the block (ghost zone included: it only needs the block's own coefficient values)
                self.ops.var_rdiag(d.a[l], d.rd[l], b.lnx, b.lny, self.h[l][0], self.h[l][1])
        NXa, NYa = self.shapes[self.Ld]
        self.ops.coarse_coefficient(np.ascontiguousarray(a_at(np.arange(NXa) << self.Ld, np.arange(NYa) << self.Ld), dtype=np.float64))
        self.var = True
        self._last_norm_parts = None
        self._norm_value = None

    def set_problem(self, rhs_of_block, u0_of_block=None):
        """rhs_of_block(block) -> (lnx, lny) array of f on that block (ghost zone and boundary included)."""
        torch = self.torch
        self._settle()
        for d in self.doms.values():
            b = d.blk[0]
            d.rhs[0][:b.lnx, :b.lny] = torch.as_tensor(np.ascontiguousarray(rhs_of_block(b), dtype=self.ldt[0])).to(d.rhs[0].device)
            d.u[0].zero_()
            if u0_of_block is not None:
                d.u[0][:b.lnx, :b.lny] = torch.as_tensor(np.ascontiguousarray(u0_of_block(b), dtype=self.ldt[0])).to(d.u[0].device)
            if d.t[0] is not None:
                d.t[0].copy_(d.u[0])
            if d.s[0] is not None:
                d.s[0].copy_(d.u[0])
            if self.mode == "fused":
                # boundary ring of every coarse rhs = injected ring of f (r = f on boundary cells), once per rhs;
                # sum of f^2 over the physical boundary cells of the exclusive window, for the norm
                for l in range(self.Ld):
                    bf, bc = d.blk[l], d.blk[l + 1]
                    ci, cj = bf.coarse_offsets(bc)
                    target = d.rc if l + 1 == self.Ld else d.rhs[l + 1]
                    self.ops.inject_ring(d.rhs[l], target, bf.lnx, bf.lny, bc.lnx, bc.lny, bf.sides, ci, cj)
                ring = torch.zeros(1, dtype=torch.float64, device=d.rhs[0].device)
                if b.sides & SIDE_ILO:
                    ring = ring + self.ops.sumsq(d.rhs[0], 0, 1, b.j_lo, b.j_hi)
                if b.sides & SIDE_IHI:
                    ring = ring + self.ops.sumsq(d.rhs[0], b.lnx - 1, b.lnx, b.j_lo, b.j_hi)
                if b.sides & SIDE_JLO:
                    ring = ring + self.ops.sumsq(d.rhs[0], max(b.i_lo, 1), min(b.i_hi, b.lnx - 1), 0, 1)
                if b.sides & SIDE_JHI:
                    ring = ring + self.ops.sumsq(d.rhs[0], max(b.i_lo, 1), min(b.i_hi, b.lnx - 1), b.lny - 1, b.lny)
                if d.ring_sumsq is None:
                    d.ring_sumsq = ring
                else:
                    d.ring_sumsq.copy_(ring)           # same buffer: a recorded plan holds its pointer
        if self.mode == "fused" and self.Ld > 0 and getattr(self.ops, "plan_capable", False):
            # the boundary ring of the gathered coarse rhs is final now (its interior is rewritten every cycle): the
            # replicated engine takes it -- and the rings of its own coarser levels -- once per problem
            self._gather_coarse_rhs()
            self.ops.coarse_begin(self.rhs_a, same_ring=False)
        self._last_norm_parts = None
        self._norm_value = None

    def residual_norm(self):
        hx, hy = self.h[0]
        if self._norm_pending:
            self._norm_value = (self._norm_plan or self._plan_back).wait()
            self._norm_pending = False
        if self._norm_value is not None:            # a native cycle brought the sum back with it
            return math.sqrt(hx * hy * self._norm_value)
        if self._last_norm_parts is not None:       # the up leg of the last cycle already summed r^2 over the owned cells
            return math.sqrt(hx * hy * self.allreduce_sum(self._last_norm_parts))
        if self.mode == "fused":
            self.exchange("u", 0)
        parts = {}
        if self.var:
            # no stand-alone variable-coefficient residual on device arrays: an up leg without sweeps on a zero correction
            # (out = u + P 0 = u, then sum r^2 over the owned cells)
            for r, d in self.doms.items():
                b, bc = d.blk[0], d.blk[1]
                ci, cj = b.coarse_offsets(bc)
                if d.zc is None:
                    d.zc = self.ops.alloc(bc.lnx, bc.lny, self.ldt[1])
                win = (max(b.i_lo, 1), min(b.i_hi, b.lnx - 1), max(b.j_lo, 1), min(b.j_hi, b.lny - 1))
                res = self.ops.up_leg(self.smk, d.u[0], d.rhs[0], d.t[0], d.zc, b.lnx, b.lny, bc.lnx, bc.lny, ci, cj, b.sides, hx, hy,
                                      self.omega, self.coeff, 0, (b.gx0 + b.gy0) & 1, win, acoef=d.a[0], rdiag=d.rd[0])
                d.t[0].copy_(d.u[0])
                parts[r] = res + d.ring_sumsq
            return math.sqrt(hx * hy * self.allreduce_sum(parts))
        for r, d in self.doms.items():
            b = d.blk[0]
            tmp = d.r[0] if d.r[0] is not None else d.t[0]
            self.ops.residual(d.u[0], d.rhs[0], tmp, b.lnx, b.lny, hx, hy, self.coeff)
            parts[r] = self.ops.sumsq(tmp, b.i_lo, b.i_hi, b.j_lo, b.j_hi)
            if d.r[0] is None:                       # t doubled as scratch: restore its boundary ring / contents
                tmp.copy_(d.u[0])
        return math.sqrt(hx * hy * self.allreduce_sum(parts))

    def iterate_sumsq(self):
        """sum of u^2 over the whole grid (every rank its exclusive window), as a Python float on every rank"""
        self._settle()
        parts = {}
        for r, d in self.doms.items():
            b = d.blk[0]
            parts[r] = self.ops.sumsq(d.u[0], b.i_lo, b.i_hi, b.j_lo, b.j_hi)
        total = None
        for r in self.ranks:
            total = parts[r] if total is None else total + parts[r]
        if self.dist is not None:
            self.dist.all_reduce(total)
        return float(total.item())

    def take_iterate_from(self, other):
        """The fine iterate of `other` (same decomposition, another working precision) becomes this solver's iterate:
        the on-device cast of PrecisionManager.convert_array (core/precision.py:106-134), ghost zone included."""
        self._settle()
        for r, d in self.doms.items():
            b = d.blk[0]
            d.u[0][:b.lnx, :b.lny].copy_(other.doms[r].u[0][:b.lnx, :b.lny])      # torch casts on the device; pitches differ
            if d.t[0] is not None:
                # the ping-pong partner only needs the outermost ring (Dirichlet values on physical edges; the legs rewrite
                # everything inside it)
                u, t = d.u[0], d.t[0]
                for t in (d.t[0], d.s[0]):
                    if t is not None:
                        t[0, :b.lny].copy_(u[0, :b.lny]); t[b.lnx - 1, :b.lny].copy_(u[b.lnx - 1, :b.lny])
                        t[:b.lnx, 0].copy_(u[:b.lnx, 0]); t[:b.lnx, b.lny - 1].copy_(u[:b.lnx, b.lny - 1])
        self._last_norm_parts = None
        self._norm_value = None

    def local_solution(self, rank):
        d = self.doms[rank]
        b = d.blk[0]
        return b, d.u[0][:b.lnx, :b.lny].cpu().numpy()

    def close(self):
        self._drop_plan()
        self._comm = None                        # shared per process: dist_plan.shutdown() destroys it
        self.ops.close()


def sine_rhs_block(b, domain=(0.0, 1.0, 0.0, 1.0)):
    """f = 2 pi^2 sin(pi x) sin(pi y) on one block, from GLOBAL indices (identical bits on every rank count)."""
    hx, hy = (domain[1] - domain[0]) / (b.NX - 1), (domain[3] - domain[2]) / (b.NY - 1)
    x = np.linspace(domain[0], domain[1], b.NX)[b.gx0:b.gx0 + b.lnx]
    y = np.linspace(domain[2], domain[3], b.NY)[b.gy0:b.gy0 + b.lny]
    del hx, hy
    return 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]


# ------------------------------------------------------------------------------------------------
# bench.py --gpus N (N > 1): weak scaling, 4097^2 points per GPU
# ------------------------------------------------------------------------------------------------
def stagnating(hist):
    """should_promote_precision on the last five residual norms (core/precision.py:189-246; csrc/mghip.hip: stagnating)."""
    if len(hist) < 5:
        return False
    r = hist[-5:]
    ratios = [r[i] / r[i - 1] for i in range(1, 5) if r[i - 1] > 0]
    if ratios:
        if sum(ratios) / len(ratios) > 0.9:
            return True
        rel = [abs(r[i] - r[i - 1]) / r[i - 1] for i in range(1, 5) if r[i - 1] > 0]
        if rel and sum(rel) / len(rel) < 1e-3:
            return True
    return all(r[i] >= r[i - 1] * 0.99 for i in range(1, 5))


def fp32_phase_pays(hx, hy, domain, coeff=-1.0, sigma=0.0):
    """csrc/mghip.hip fp32_phase_pays: the fp32 residual floor relative to ||r_0|| is at most eps32 diag(A) / lambda_min (a
    property of the grid); with a contraction of ~0.15 per cycle the fp32 phase is good for log(that) / log(0.15) cycles, and
    it is entered only when that is at least two -- its switches cost about one cycle's saving."""
    lx, ly = domain[1] - domain[0], domain[3] - domain[2]
    lam = abs(coeff) * math.pi**2 * (1.0 / (lx * lx) + 1.0 / (ly * ly)) + sigma
    ratio = 2.0**-24 * (2.0 / (hx * hx) + 2.0 / (hy * hy) + sigma) / lam
    return ratio > 0.0 and math.log(ratio) / math.log(0.15) >= 2.0


class AdaptivePolicy:
    """The engine's adaptive rule (csrc/mghip.hip: adapt, one-way variant of core/precision.py:270-302) as host logic
    for drivers that hold one solver per precision: start in double, drop to single on a large first residual,
    promote for good when ||r|| < 10 thr or the fp32 iteration stagnates."""

    EPS32 = 2.0 ** -24

    def __init__(self, thr, fp32_pays=True):
        """fp32_pays: the fp32 phase is good for at least two cycles on this grid (fp32_phase_pays); False: stay in double"""
        self.thr, self.phase, self.promoted, self.hist = thr, "f64", False, []
        self.fp32_pays = bool(fp32_pays)
        self.floor = 0.0          # eps32 * diag(A) * ||u||_h: the residual an fp32 iterate can reach (set_floor; 0: not evaluated)
        self.reason = None        # why the fp32 phase ended: "threshold" / "stagnation" / "fp32_floor"; "fp32_skipped": never begun

    def set_floor(self, diag, u_norm_h):
        """after the first fp32 cycle (csrc/mghip.hip, iterate_impl): within a factor 2 of this floor another fp32 cycle
        cannot lower the residual, and the policy promotes at once instead of waiting for the stagnation window to fill"""
        self.floor = self.EPS32 * diag * u_norm_h

    def floor_due(self):
        return self.phase == "f32" and not self.promoted and self.floor == 0.0

    def before_cycle(self, rn):
        """-> the precision the coming cycle runs in (the caller moves the iterate when it differs from .phase)"""
        want = self.phase
        if not self.promoted:
            if self.phase == "f64" and rn > 100.0 * self.thr and not self.hist:
                if self.fp32_pays:
                    want = "f32"
                else:
                    self.reason = "fp32_skipped"
            elif self.phase == "f32" and (rn < 10.0 * self.thr or stagnating(self.hist) or 0.0 < self.floor and rn <= 2.0 * self.floor):
                want, self.promoted = "f64", True
                self.reason = "threshold" if rn < 10.0 * self.thr else ("stagnation" if stagnating(self.hist) else "fp32_floor")
        if want != self.phase:
            self.phase = want
            self.hist = []
        return want

    def after_cycle(self, rn):
        self.hist.append(rn)

    def switch_likely(self):
        """Will the norm of the cycle about to run change the precision?  Extrapolated from the last two norms of this
        phase, as the engine does before it queues a speculative front part (csrc/mghip.hip, iterate_impl)."""
        if self.floor_due():
            return True                # the floor is evaluated from the iterate the coming cycle leaves and usually ends the phase
        if self.promoted or self.phase != "f32" or len(self.hist) < 2:
            return False
        prev, last = self.hist[-2], self.hist[-1]
        guess = last * (min(1.0, last / prev) if prev > 0 else 1.0)
        return guess < 10.0 * self.thr or stagnating(self.hist + [guess])


class FixedPolicy:
    """One working precision for the whole solve (the interface of AdaptivePolicy)."""

    def __init__(self, name):
        self.phase, self.promoted, self.hist, self.reason = name, True, [], None

    def before_cycle(self, rn):
        return self.phase

    def after_cycle(self, rn):
        self.hist.append(rn)

    def switch_likely(self):
        return False


class DecomposedSolve:
    """The loop of mg_iterate (csrc/mghip.hip; solvers/multigrid.py:219-246) on the decomposed hierarchy: policy check ->
    cycle -> ||r|| -> record -> absolute stop test, driving one DistributedMultigrid per working precision (they share the
    decomposition; the iterate moves between them with take_iterate_from, the on-device cast of
    PrecisionManager.convert_array).  bench.py --gpus N and DistributedMultigridSolver.solve both run THIS loop.

    solvers: {"f64": DistributedMultigrid, "f32": ...} (one entry for a fixed precision);
    policy:  "fixed" or "adaptive" (AdaptivePolicy with `switch_threshold`)."""

    def __init__(self, solvers, policy="fixed", switch_threshold=1e-6):
        self.solvers = dict(solvers)
        if policy not in ("fixed", "adaptive"):
            raise ValueError(f"Unknown precision policy: {policy}")
        if policy == "adaptive" and set(self.solvers) != {"f32", "f64"}:
            raise ValueError("the adaptive policy switches between an 'f32' and an 'f64' solver")
        self.policy_kind, self.thr = policy, switch_threshold
        self.start = "f64" if "f64" in self.solvers else next(iter(self.solvers))
        self.policy = None
        self.rn = None
        self.switches = 0

    def _new_policy(self):
        if self.policy_kind != "adaptive":
            return FixedPolicy(self.start)
        sv = self.solvers[self.start]
        return AdaptivePolicy(self.thr, fp32_phase_pays(sv.h[0][0], sv.h[0][1], sv.domain, sv.coeff))

    def set_problem(self, rhs_of_block, u0_of_block=None):
        """every precision takes the right-hand side (and the initial guess); a solve starts in `start` (double when there
        is a choice: PrecisionManager's default precision, core/precision.py:26-45).  Returns the initial residual norm."""
        for sv in self.solvers.values():
            sv.set_problem(rhs_of_block, u0_of_block)
        self.policy = self._new_policy()
        self.switches = 0
        self.rn = self.solvers[self.start].residual_norm()
        return self.rn

    @property
    def current(self):
        """the solver that holds the iterate"""
        return self.solvers[self.policy.phase]

    def step(self, tol=0.0):
        """policy check (before the cycle, solvers/multigrid.py:224-227) -> cycle -> norm; returns the new norm"""
        policy, solvers = self.policy, self.solvers
        had = policy.phase
        now = policy.before_cycle(self.rn)
        if now != had:
            solvers[now].take_iterate_from(solvers[had])
            self.switches += 1
        sv = solvers[now]
        # no speculative front part across a precision switch the policy can see coming, nor across the end of the solve
        # (it would run and be dropped): the norm in flight extrapolated from the last two, as iterate_impl does
        ends = False
        if tol > 0.0 and len(policy.hist) >= 2 and policy.hist[-2] > 0:
            ends = policy.hist[-1] * min(1.0, policy.hist[-1] / policy.hist[-2]) < tol
        sv.speculate = not (policy.switch_likely() or ends)
        sv.cycle(0)
        self.rn = sv.residual_norm()
        policy.after_cycle(self.rn)
        if getattr(policy, "floor_due", None) is not None and policy.floor_due():
            hx, hy = sv.h[0]
            policy.set_floor(2.0 / (hx * hx) + 2.0 / (hy * hy), math.sqrt(hx * hy * sv.iterate_sumsq()))
        return self.rn

    def run(self, tol, max_iterations):
        """-> (history, phase per cycle, converged): cycles until ||r|| < tol (absolute, solvers/base.py:134)"""
        hist, phases = [], []
        converged = False
        for _ in range(max_iterations):
            rn = self.step(tol)
            hist.append(rn)
            phases.append(self.policy.phase)
            if rn < tol:
                converged = True
                break
        return hist, phases, converged

    def close(self):
        for sv in self.solvers.values():
            sv.close()


def _first_difference(a, b):
    """(i, j) of the first element where two equally shaped tensors differ bitwise, or None"""
    import torch
    ne = (a.view(torch.int64 if a.element_size() == 8 else torch.int32) != b.view(torch.int64 if b.element_size() == 8 else torch.int32))
    idx = torch.nonzero(ne)
    return None if idx.numel() == 0 else tuple(int(v) for v in idx[0])


def plan_selfcheck(sv, set_problem, dist):
    """Before the clock: one cycle through the Python driver and the same cycle replayed from the recorded plan, from the
    same start, must leave the same iterate BIT FOR BIT on every rank and the same norm.  Returns a dict for the bench line;
    mismatch = {"rank", "first_diff", ...} on the ranks that differ (the caller aborts non-zero)."""
    torch = sv.torch
    if not sv.native:
        return {"ran": False, "reason": "python driver only (no native plan on this backend / mode)"}
    (r, d), = sv.doms.items()
    b = d.blk[0]
    was = sv.native
    set_problem(sv)
    sv.native = False
    sv.cycle(0)
    n_py = sv.residual_norm()
    u_py = d.u[0][:b.lnx, :b.lny].clone()
    sv.native = was
    set_problem(sv)
    sv.cycle(0)                       # records (first time) or replays
    sv.residual_norm()
    if sv.native:                     # the recording did not fall back: this one is a replay for certain
        set_problem(sv)
        replays_before = sv.native_cycles
        sv.cycle(0)
        n_na = sv.residual_norm()
        replayed = sv.native_cycles == replays_before + 1
        u_na = d.u[0][:b.lnx, :b.lny]
        diff = _first_difference(u_py, u_na)
        bad = (diff is not None) or not (n_py == n_na)
    else:
        replayed, diff, bad, n_na = False, None, False, n_py
    flag = torch.tensor([1 if bad else 0], dtype=torch.int32, device=u_py.device)
    if dist is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    res = {"ran": True, "replayed": bool(replayed), "bit_identical": not bool(int(flag.item())), "norm_python": n_py, "norm_native": n_na}
    if bad:
        res["mismatch"] = {"rank": r, "block": [b.gx0, b.gy0, b.lnx, b.lny], "first_diff": diff}
    return res


def bench_main(args, rank, local_rank, world):
    """bench.py --gpus N (N > 1): BASELINE config 3's workload per GPU (4097^2, adaptive fp32 -> fp64, V(2,2) weighted
    Jacobi) on a px x py block decomposition -- weak scaling of the N = 1 bench line.  The precision policy is the
    engine's (core/precision.py:270-302 with the one-way promotion): start in double, drop to single while
    ||r|| > 100 thr, promote for good once ||r|| < 10 thr; it switches between two solvers that share the decomposition
    (DecomposedSolve: the loop DistributedMultigridSolver.solve runs too).

    Before the clock starts the run checks itself: every rank of the communicator is counted (`ranks_seen`), one cycle is
    run through the Python driver and replayed from the recorded plan from the same start and the two iterates are compared
    bit for bit (`selfcheck`; a mismatch prints the first differing block and exits non-zero).  After the timed region three
    diagnostic cycles are bracketed with timing events per phase (`phases_ms_per_cycle`: legs, halo copies, send/recv groups
    incl. the wait for the peers, coarse all-gather, replicated engine, all-reduce).  A rank whose plan times out
    (MG_PLAN_TIMEOUT_S) reports and leaves with os._exit -- it never synchronises on the stuck streams again.

    Test hook (tests/test_distributed_cpu.py): MG_DIST_BACKEND=gloo with MG_BENCH_OPS=module:Class runs the same driver
    on CPU tensors with a stand-in kernel provider; without it the kernels are libmghip's and a GPU is required."""
    import importlib
    import sys
    import torch
    import torch.distributed as dist
    if world != args.gpus:
        raise RuntimeError(f"bench.py --gpus {args.gpus} under a launcher that started {world} ranks (WORLD_SIZE)")
    # rehearsal knobs (one-GPU box): MG_DIST_BACKEND=gloo MG_DIST_SAME_DEVICE=1 runs every rank on cuda:0 over gloo
    backend = os.environ.get("MG_DIST_BACKEND", "nccl")
    ops_spec = os.environ.get("MG_BENCH_OPS") if backend == "gloo" else None
    on_gpu = ops_spec is None
    if os.environ.get("MG_DIST_SAME_DEVICE") == "1":
        local_rank = 0
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py --gpus N needs MI355X devices (no CPU fallback)"
        torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    try:
        return _bench_ranks(args, rank, local_rank, world, backend, ops_spec, on_gpu, torch, dist, importlib)
    except _lib.PlanTimeout as exc:
        # the RCCL work of the stuck cycle is still queued: any synchronisation (torch.cuda.synchronize, mg_destroy, hipFree,
        # destroy_process_group) would hang on it.  Report and leave; the launcher tears the other ranks down.
        sys.stderr.write(f"bench.py rank {rank}: {exc}\n")
        sys.stderr.flush()
        os._exit(3)


def _bench_ranks(args, rank, local_rank, world, backend, ops_spec, on_gpu, torch, dist, importlib):
    import sys
    px, py = process_grid(world)
    m = args.n - 1
    NX, NY = px * m + 1, py * m + 1
    # unit cells: the domain grows with the process grid so that hx = hy = 1/(n-1) as on one GPU
    domain = (0.0, float(px), 0.0, float(py))
    thr = 1e-6                                                    # BASELINE config 3: switch_threshold
    if on_gpu:
        dev = torch.device("cuda", local_rank)
        providers = (("f32", HipOps(np.float32, dev, managed_single=True)), ("f64", HipOps(np.float64, dev)))
    else:
        modname, cls = ops_spec.split(":")
        factory = getattr(importlib.import_module(modname), cls)
        providers = (("f32", factory(np.float32)), ("f64", factory(np.float64)))
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    solvers = {}
    # MG_DIST_NATIVE=0: the Python driver every cycle (default: recorded cycle plans wherever they apply)
    native = "auto" if os.environ.get("MG_DIST_NATIVE", "1") != "0" else False
    for name, ops in providers:
        solvers[name] = DistributedMultigrid(NX, NY, px, py, [rank], ops, dist, domain=domain, smoother="jacobi", omega=0.8,
                                             cycle="V", pre=2, post=2, agglomerate_at=getattr(args, "agglomerate_at", 1025),
                                             native=native)
    loop = DecomposedSolve(solvers, "adaptive", thr)
    rhs_of = lambda b: sine_rhs_block(b, domain)

    # ---- who is here: every rank adds one, over torch.distributed and (native plans) over the library's own communicator ----
    seen = torch.ones(1, dtype=torch.int32, device="cuda" if (on_gpu and backend == "nccl") else "cpu")
    dist.all_reduce(seen)
    ranks_seen = int(seen.item())
    who = [None] * world
    dist.all_gather_object(who, {"rank": rank, "device": (torch.cuda.current_device() if on_gpu else "cpu"), "pid": os.getpid()})

    # ---- untimed set-up: the plan self-check builds both cycle plans (and the library's RCCL communicator) -------------------
    checks = {}
    for name, sv in solvers.items():
        checks[name] = plan_selfcheck(sv, lambda s: s.set_problem(rhs_of), dist)
    failed = [c for c in checks.values() if c.get("ran") and not c["bit_identical"]]
    if failed:
        for name, c in checks.items():
            if "mismatch" in c:
                sys.stderr.write(f"bench.py rank {rank}: native replay != Python driver ({name}): {json.dumps(c['mismatch'])}\n")
        sys.stderr.flush()
        sync()
        dist.barrier()
        for x in solvers.values():
            x.close()
        dist.destroy_process_group()
        return 4
    comm_ranks = None
    for sv in solvers.values():
        if sv._comm is not None:
            comm_ranks = sv._comm.ranks()[0]
    for sv in solvers.values():           # the Python-driver fallback needs its warm-up too
        if not sv.native:
            sv.set_problem(rhs_of)
            sv.cycle(0)
            sv.residual_norm()

    K, W = args.steps, args.warmup
    loop.set_problem(rhs_of)
    for _ in range(W):
        loop.step()
    r0 = loop.set_problem(rhs_of)
    hist, phases = [], []
    for sv in solvers.values():
        sv.exchanges = 0
        sv.native_cycles = 0
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(K):
        hist.append(loop.step())
        phases.append(loop.policy.phase)
    sync()
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if (on_gpu and backend == "nccl") else "cpu")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    exchanges = sum(x.exchanges for x in solvers.values()) / max(1, K)
    native_cycles = sum(x.native_cycles for x in solvers.values())
    # iterations to tolerance / the plateau of the reference's absolute norm (untimed continuation of the same solve)
    long_hist = list(hist)
    for _ in range(max(0, 40 - K)):
        long_hist.append(loop.step())
    tail = sorted(long_hist[-5:])
    floor = tail[len(tail) // 2]
    first = lambda vals, t: next((k + 1 for k, v in enumerate(vals) if v < t), None)
    # ---- per-phase device times: three more cycles of the dominant precision, every phase bracketed by timing events ------
    dom = "f64" if phases.count("f64") >= phases.count("f32") else "f32"
    sv = solvers[dom]
    sv.profile_phases(True)
    ncyc = 3
    for _ in range(ncyc):
        sv.cycle(0)
        sv.residual_norm()
    ph = sv.collect_phase_times()
    sv.profile_phases(False)
    phases_ms = {k: v / ncyc for k, v in ph.items()}
    # roofline leg (rank 0): the dominant kernel of the timed region -- the level-0 up leg (prolongation + 2 sweeps +
    # norm) of the precision that ran most cycles -- on this rank's block, timed with events on its own stream
    sv._settle()
    d0 = sv.doms[rank]
    b0, b1 = d0.blk[0], d0.blk[1]
    hx0, hy0 = sv.h[0]
    ci, cj = b0.coarse_offsets(b1)
    e = d0.ec if sv.Ld == 1 else d0.u[1]
    win = (max(b0.i_lo, 1), min(b0.i_hi, b0.lnx - 1), max(b0.j_lo, 1), min(b0.j_hi, b0.lny - 1))

    def leg():
        sv.ops.up_leg(sv.smk, d0.u[0], d0.rhs[0], d0.t[0], e, b0.lnx, b0.lny, b1.lnx, b1.lny, ci, cj, b0.sides, hx0, hy0,
                      sv.omega, sv.coeff, sv.post, (b0.gx0 + b0.gy0) & 1, win)
    reps = 20 if on_gpu else 1
    leg()
    if on_gpu:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            leg()
        ev1.record()
        torch.cuda.synchronize()
        ms_leg = ev0.elapsed_time(ev1) / reps
    else:
        t1 = time.perf_counter()
        leg()
        ms_leg = (time.perf_counter() - t1) * 1e3
    w = 8 if dom == "f64" else 4
    moved, unfused = 3.25 * w * b0.lnx * b0.lny, 10.25 * w * b0.lnx * b0.lny
    if rank == 0:
        value = NX * NY * K / dt / 1e6
        s0 = solvers["f64"]
        gbs = moved / (ms_leg * 1e-3) / 1e9
        print(json.dumps({
            "metric": "MDoF/s per V-cycle on 2D Poisson", "value": value, "unit": "MDoF/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32->f64 (adaptive)", "data": "synthetic",
            "config": {"workload": f"2D Poisson {NX}x{NY} adaptive fp32->fp64 (switch_threshold={thr:g}), V(2,2) weighted-Jacobi "
                                   f"omega=0.8, {px}x{py} block decomposition ({args.n}^2 per GPU), RCCL halo exchange ({s0.mode} legs, "
                                   f"ghost width {s0.G}), {s0.L} levels ({s0.Ld} distributed, rest replicated after all-gather)",
                       "grid": [NX, NY], "levels": s0.L, "cycle": "V(2,2)", "smoother": "jacobi",
                       "parallelism": f"dd{px}x{py}", "backend": backend if on_gpu else f"{backend} (CPU rehearsal, {ops_spec})"},
            "cycles_fp32": phases.count("f32"), "cycles_fp64": phases.count("f64"),
            "residual_initial": r0, "residual_first": hist[0], "residual_last": hist[-1],
            "iterations": K, "iterations_to_1e-10_absolute": first(long_hist, 1e-10),
            "iterations_to_1e-10_relative": first([v / r0 for v in long_hist], 1e-10),
            "iterations_to_1e-9_absolute": first(long_hist, 1e-9),
            "residual_floor": floor, "iterations_to_floor": first(long_hist, 2.0 * floor),
            "roofline": {"bound": "hbm", "kernel": f"fused up leg ({'rb_leg_kernel' if b0.lnx * b0.lny > 1100 * 1100 else 'fused_jacobi_kernel'}) {dom} on the local {b0.lnx}x{b0.lny} block (rank 0, level 0)",
                         "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None, "launch_ms": ms_leg,
                         "bytes_per_launch": moved, "unfused_equivalent_bytes": unfused,
                         "unfused_equivalent_gbs": unfused / (ms_leg * 1e-3) / 1e9,
                         "note": "achieved = bytes the launch must move (3.25 words per cell of the local block, ghost zone "
                                 "included) / launch time; unfused_equivalent_* prices the same work as one launch per operator "
                                 "(SURVEY 8d)"},
            "exchanges_per_cycle": exchanges,
            "ranks_seen": ranks_seen, "rank_devices": who,
            "driver": {"native_plan_cycles": native_cycles, "python_cycles": K - native_cycles,
                       "fallback": next((x.native_failure for x in solvers.values() if x.native_failure), None),
                       "rccl_comm_ranks": comm_ranks,
                       # level-0 up legs of cycle k + down legs of cycle k + 1 as one launch per block (MG_DIST_SPAN=0: two)
                       "spanning_scheme": {name: bool(x._span_usable()) for name, x in solvers.items()},
                       "rccl_multi_rank_replay": ("exercised in this run" if (native_cycles > 0 and world > 1 and backend == "nccl") else
                                                  "not exercised (no multi-rank RCCL plan ran here)"),
                       "selfcheck": checks},
            "phases_ms_per_cycle": dict(phases_ms, precision=dom, cycles=ncyc, rank=0,
                                        source=("timing events inside mg_plan_run" if sv.native else
                                                ("timing events around the Python driver's phases" if on_gpu else "wall clock (CPU rehearsal)"))),
            "note": "distributed levels: communication-avoiding fused legs (two launches and about one halo exchange per "
                    "level and cycle); a cycle is recorded once through the Python driver and then replayed from C++ -- one "
                    "mg_plan_run per cycle enqueues the kernels, the RCCL send/recv groups, the coarse all-gather and the "
                    "norm all-reduce on two HIP streams (MG_DIST_NATIVE=0: torch.distributed P2P from Python every cycle); "
                    "the replicated coarse hierarchy runs on the fused single-GPU engine; same precision policy as the "
                    "N = 1 line; multi-rank RCCL replay has never run before the first real multi-GPU run: `driver` says which "
                    "path this run took and `selfcheck` that replay and Python driver agreed bit for bit before the clock",
        }), flush=True)
    for x in solvers.values():
        x.close()
    if on_gpu:
        from . import dist_plan
        dist_plan.shutdown()
    dist.destroy_process_group()
    return 0
