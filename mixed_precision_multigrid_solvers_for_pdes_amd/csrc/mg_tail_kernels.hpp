// Register-resident coarse tail: every level of at most 65^2 cells, down to and including the 5 x 5 coarsest solve, in ONE
// workgroup per visit -- the sub-cycle a W- or F-cycle enters 2^l times per fine-grid cycle.
//
// coarse_tail_kernel (mg_kernels.hpp) keeps these levels in LDS and pays ~0.4 us per barrier-separated stage: every cell
// update is five LDS reads behind index arithmetic that runs on run-time level geometry.  Here the levels are the squares
// 2^m + 1 every dyadic hierarchy ends in -- 65, 33, 17, 9, 5 -- with COMPILE-TIME geometry, and the iterate and the
// right-hand side of every level live in REGISTERS for the whole visit:
//
//   * a level of n = NC + 1 points per side is the NC x NC cells (i, j), 0 <= i, j < NC: row 0 and column 0 are boundary
//     cells (they hold 0 and are never updated), row / column NC is the far boundary (never stored).  A lane owns 4
//     consecutive rows of one column ("strip" s = rows 4s .. 4s + 3); a wave holds 64 / NC strips side by side (lane =
//     sub-strip * NC + column): 65^2 is 16 waves, 33^2 four, 17^2 and 9^2 one;
//   * vertical neighbours: the lane's own rows; across strips the first / last row of every strip crosses a small LDS
//     exchange buffer once per stage (1 write pair, 1 sync, 1 read pair; double-buffered by stage parity);
//   * lateral neighbours: DPP wave_shr:1 / wave_shl:1.  Sub-strips sit side by side in a wave, so the lane left of a
//     sub-strip's column 1 is its own column 0 and the lane right of column NC - 1 is the NEXT sub-strip's column 0 (or
//     nothing: bound_ctrl zero) -- a boundary cell, value 0, exactly what the stencil needs there;
//   * full weighting in the fine layout (rows k -+ 1 of the residual strip, DPP for west / east), coarse values staged
//     through LDS into the coarse level's lanes; bilinear interpolation from the coarse iterate staged in LDS (n x n with
//     its zero ring, so the far-edge cells read zeros);
//   * levels of one wave (17^2, 9^2, 5^2) synchronise with wave-level fences only: the other waves wait at the next
//     workgroup barrier of the level above;
//   * the 5 x 5 solve is lexgs_5x5_zero_ring (mg_kernels.hpp: the reference's lexicographic Gauss-Seidel to coarse_tol,
//     bit for bit) or, mg_config.coarse_direct, u = A^-1 f.
//
// Arithmetic per cell: the expressions of the single-operator kernels in the same order (solvers/smoothers.py:62-84,
// 175-207; operators/laplacian.py:73-77; operators/transfer.py:100-124, 234-267) -- results are bit-identical to
// coarse_tail_kernel and to one launch per operator.
#pragma once

#include "mg_kernels.hpp"
#include "mg_rb_kernels.hpp"

namespace mg {

constexpr int kT2MaxLev = 4;           // register levels (65, 33, 17, 9); the 5 x 5 level is the coarsest solve

struct Tail2Level {
  double ihx2, ihy2, invD, diag;
  int use_div;                          // 1: divide by the diagonal (1/D not exact)
};
struct Tail2Args {
  int ld_top, maxit, pre, post, colour_offset, direct;
  int reps[kT2MaxLev];                  // recursion count below register level i (V 1, W 2, F 2^(L - l - 2))
  double omega, coeff, tol_x, sigma;
  Tail2Level lv[kT2MaxLev];             // register level i = 0 (top) ...
  double hx2_5, hy2_5, diag_5, hxhy_5;  // the 5 x 5 level
  int exact_5;                          // hx^2, hy^2 and the diagonal are powers of two there
  const void* ring5;                    // the coarsest level's rhs array in HBM (dtype TCO): its boundary ring is the injected
  int ring5_ld;                         //   ring of f, which the reference's stop test counts (solvers/base.py:271-283)
  double minv[81];                      // direct: inverse of the 9 x 9 coarsest matrix, row-major
};

template <int NC> struct T2Geo {
  static constexpr int N = NC + 1;              // points per side
  static constexpr int SPW = 64 / NC;           // strips per wave
  static constexpr int NS = NC / 4;             // strips of 4 rows
  static constexpr int WAVES = (NS + SPW - 1) / SPW;
  static constexpr bool BLOCK = WAVES > 1;      // more than one wave: workgroup barriers; else wave-level fences
  static constexpr int XELEMS = NS * 2 * NC;    // one exchange buffer (elements)
};

template <typename T> struct T2Lane {           // where this lane sits on a level
  int col, strip;
  bool active;                                  // the lane holds cells of this level
  bool colok;                                   // col >= 1 (column 0 is boundary)
};
template <typename T, int NC> __device__ __forceinline__ T2Lane<T> t2_lane() {
  using G = T2Geo<NC>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T2Lane<T> p;
  p.col = lane & (NC - 1);
  p.strip = wave * G::SPW + lane / NC;
  p.active = p.strip < G::NS;
  p.colok = p.active && p.col >= 1;
  return p;
}

template <bool BLOCK> __device__ __forceinline__ void t2_sync() {
  if (BLOCK) __syncthreads();
  else wave_lds_fence<int>();
}

// first / last row of every strip through LDS: `above` = last row of the strip above, `below` = first row of the strip below
template <typename T, int NC>
__device__ __forceinline__ void t2_exchange(T* __restrict__ xb, const T2Lane<T>& p, T top, T bottom, T& above, T& below) {
  using G = T2Geo<NC>;
  if (p.active) {
    xb[(p.strip * 2 + 0) * NC + p.col] = top;
    xb[(p.strip * 2 + 1) * NC + p.col] = bottom;
  }
  t2_sync<G::BLOCK>();
  above = (p.active && p.strip > 0) ? xb[((p.strip - 1) * 2 + 1) * NC + p.col] : T(0);
  below = (p.active && p.strip < G::NS - 1) ? xb[((p.strip + 1) * 2 + 0) * NC + p.col] : T(0);
}

template <typename T> struct T2Const { T ihx2, ihy2, invD, D, omega, one_m_omega, coeff; bool use_div; int coff; };

// one weighted-Jacobi sweep (SM = kSmJacobi) or one colour pass of red-black GS (kSmRbgs, colour 0 = (i + j) even first)
template <typename T, int NC, int SM>
__device__ __forceinline__ void t2_pass(T (&U)[4], const T (&F)[4], T* __restrict__ xbuf, int& stage, const T2Lane<T>& p,
                                        const T2Const<T>& c, int colour) {
  using G = T2Geo<NC>;
  T above, below;
  t2_exchange<T, NC>(xbuf + (stage & 1) * G::XELEMS, p, U[0], U[3], above, below);
  ++stage;
  const int par0 = (4 * p.strip + p.col + c.coff) & 1;         // parity of (i + j + colour_offset) of the lane's first cell
  T prev = above;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const T mid = U[k];
    const T dn = (k < 3) ? U[k < 3 ? k + 1 : 0] : below;
    const T left = dpp_from_lower_lane<T>(mid);               // every lane active here
    const T right = dpp_from_upper_lane<T>(mid);
    const T nb = c.ihx2 * (dn + prev) + c.ihy2 * (right + left);
    const T un = c.use_div ? (F[k] + nb) / c.D : (F[k] + nb) * c.invD;
    const T res = c.one_m_omega * mid + c.omega * un;
    const bool mine = (SM != kSmRbgs) || (((par0 + k) & 1) == colour);
    const bool ok = p.colok && (k > 0 || p.strip > 0) && mine;
    const T o = ok ? res : mid;
    U[k] = o;
    prev = (SM == kSmRbgs) ? o : mid;           // red-black GS is in place: the neighbours it reads are not of this colour
  }
}

template <typename T, int NC, int SM>
__device__ __forceinline__ void t2_smooth(T (&U)[4], const T (&F)[4], T* __restrict__ xbuf, int& stage, const T2Lane<T>& p,
                                          const T2Const<T>& c, int nsweep) {
  for (int s = 0; s < nsweep; ++s) {
    if (SM == kSmRbgs) {
      t2_pass<T, NC, SM>(U, F, xbuf, stage, p, c, 0);
      t2_pass<T, NC, SM>(U, F, xbuf, stage, p, c, 1);
    } else {
      t2_pass<T, NC, SM>(U, F, xbuf, stage, p, c, 0);
    }
  }
}

// residual of the strip, then full weighting of the interior coarse cells on it; the coarse values go to fc (the coarse
// level's n x n staging array, dtype TX) -- the caller synchronises before the coarse lanes read them
template <typename T, typename TX, int NC>
__device__ __forceinline__ void t2_residual_restrict(const T (&U)[4], const T (&F)[4], T* __restrict__ xbuf, int& stage,
                                                     const T2Lane<T>& p, const T2Const<T>& c, TX* __restrict__ fc) {
  using G = T2Geo<NC>;
  constexpr int NCC = NC / 2 + 1;                 // coarse points per side
  T R[4];
  {
    T above, below;
    t2_exchange<T, NC>(xbuf + (stage & 1) * G::XELEMS, p, U[0], U[3], above, below);
    ++stage;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T up = (k > 0) ? U[k > 0 ? k - 1 : 0] : above;
      const T mid = U[k];
      const T dn = (k < 3) ? U[k < 3 ? k + 1 : 0] : below;
      const T left = dpp_from_lower_lane<T>(mid);
      const T right = dpp_from_upper_lane<T>(mid);
      const T au = c.coeff * (((dn + up) * c.ihx2 + (right + left) * c.ihy2) - mid * c.D);
      const bool ok = p.colok && (k > 0 || p.strip > 0);
      R[k] = ok ? (F[k] - au) : T(0);               // boundary cells: no interior coarse cell reads them
    }
  }
  {
    T above, below;
    t2_exchange<T, NC>(xbuf + (stage & 1) * G::XELEMS, p, R[0], R[3], above, below);
    ++stage;
    (void)below;
#pragma unroll
    for (int k = 0; k < 4; k += 2) {                // rows 4s and 4s + 2 are the even (coarse) rows of the strip
      const T up = (k > 0) ? R[k > 0 ? k - 1 : 0] : above;
      const T mid = R[k];
      const T dn = R[k + 1];
      const T nw = dpp_from_lower_lane<T>(up), w = dpp_from_lower_lane<T>(mid), sw = dpp_from_lower_lane<T>(dn);
      const T ne = dpp_from_upper_lane<T>(up), e = dpp_from_upper_lane<T>(mid), se = dpp_from_upper_lane<T>(dn);
      const T corners = ((nw + ne) + sw) + se;
      const T edges = ((up + dn) + w) + e;
      const T val = (T(1.0 / 16.0) * corners + T(1.0 / 8.0) * edges) + T(1.0 / 4.0) * mid;
      const int i = 4 * p.strip + k;
      if (p.colok && !(p.col & 1) && i >= 2) fc[(i >> 1) * NCC + (p.col >> 1)] = (TX)val;
    }
  }
}

// u += P e: `ec` is the coarse iterate as an n x n array (zero ring) of dtype TX; interpolation in TC (the fine GRID's
// dtype), the sum in the wider of (T, TC), rounded to T (operators/transfer.py:234-267 + solvers/multigrid.py:329)
template <typename T, typename TX, typename TC, int NC>
__device__ __forceinline__ void t2_prolong_add(T (&U)[4], const T2Lane<T>& p, const TX* __restrict__ ec) {
  using TS = typename std::conditional<(sizeof(TC) > sizeof(T)), TC, T>::type;
  constexpr int NCC = NC / 2 + 1;
  if (!p.active) return;
  const int jc = p.col >> 1, ic0 = 2 * p.strip;
  const bool jodd = p.col & 1;
  TC e0[3], e1[3];                                   // coarse rows 2s, 2s + 1, 2s + 2 at columns jc / jc + 1
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    e0[r] = (TC)ec[(ic0 + r) * NCC + jc];
    e1[r] = (TC)ec[(ic0 + r) * NCC + jc + 1];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = k >> 1;
    TC val;
    if ((k & 1) == 0) val = jodd ? TC(0.5) * (e0[r] + e1[r]) : e0[r];
    else val = jodd ? TC(0.25) * (((e0[r] + e1[r]) + e0[r + 1]) + e1[r + 1]) : TC(0.5) * (e0[r] + e0[r + 1]);
    const bool ok = p.colok && (k > 0 || p.strip > 0);
    if (ok) U[k] = (T)((TS)U[k] + (TS)val);
  }
}

// LDS layout of one launch (byte offsets into the dynamic pool), for the top level NCTOP
template <typename T, typename TCO, int NCTOP> struct T2Lds {
  static constexpr size_t al(size_t x) { return (x + 15) / 16 * 16; }
  static constexpr size_t kX = 0;                                                     // 2 exchange buffers of the top level
  static constexpr size_t kXBytes = al(2 * (size_t)T2Geo<NCTOP>::XELEMS * sizeof(T));
  // staging arrays of the coarser levels: level with NC cells per side has an (NC + 1)^2 array for its rhs (written by the
  // level above) and one for its iterate (read by the level above)
  static constexpr size_t lev_bytes(int nc) { return al((size_t)(nc + 1) * (nc + 1) * sizeof(T)); }
  static constexpr size_t off_f(int nc) {          // nc = NCTOP / 2, / 4, ... down to 8
    size_t o = kX + kXBytes;
    for (int m = NCTOP / 2; m > nc; m /= 2) o += 2 * lev_bytes(m);
    return o;
  }
  static constexpr size_t off_e(int nc) { return off_f(nc) + lev_bytes(nc); }
  static constexpr size_t kFive = off_f(4);                                           // su[25], sf[25] in TCO
  static constexpr size_t kTotal = kFive + al(2 * 25 * sizeof(TCO));
};

template <typename T, typename TCO, typename TC, int SM, int NCTOP> struct T2Ctx {
  const Tail2Args& a;
  unsigned char* pool;
  T* xbuf;
  int stage;
  int sweeps;
  double minv_row[9];            // direct: row `lane` of the inverse (lanes 0..8 of wave 0)
};

// ---- the 5 x 5 level: rhs = full weighting of the 9^2 level's residual (already in sf), solve, result in su -----------
template <typename T, typename TCO, typename TC, int SM, int NCTOP>
__device__ __forceinline__ void t2_solve5(T2Ctx<T, TCO, TC, SM, NCTOP>& cx, bool zero) {
  using L = T2Lds<T, TCO, NCTOP>;
  TCO* su = reinterpret_cast<TCO*>(cx.pool + L::kFive);
  TCO* sf = su + 25;
  const int lane = threadIdx.x & 63;
  const Tail2Args& a = cx.a;
  if (zero) {
    if (lane < 25) su[lane] = TCO(0);
    wave_lds_fence<int>();
  }
  if (a.direct) {
    // nine unknowns in lanes 0..8: u_i = sum_j minv[i][j] f_j, f_j broadcast from lane j; a fixed summation order
    const int li = lane < 9 ? lane : 0;
    const int g = (li / 3 + 1) * 5 + (li % 3) + 1;
    const double fv = (double)sf[g];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const double fj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fv), j), __builtin_amdgcn_readlane(__double2loint(fv), j));
      acc += cx.minv_row[j] * fj;
    }
    if (lane < 9) su[g] = (TCO)acc;
    wave_lds_fence<int>();
    cx.sweeps = 0;
    return;
  }
  const TCO hx2 = (TCO)a.hx2_5, hy2 = (TCO)a.hy2_5, diag = (TCO)a.diag_5, cf = (TCO)a.coeff;
  cx.sweeps = a.exact_5
      ? lexgs_5x5_zero_ring<TCO, false, true>(su, sf, hx2, hy2, diag, cf, TCO(1), TCO(0), a.hxhy_5, a.tol_x, a.maxit, lane, nullptr, TCO(0))
      : lexgs_5x5_zero_ring<TCO, false, false>(su, sf, hx2, hy2, diag, cf, TCO(1), TCO(0), a.hxhy_5, a.tol_x, a.maxit, lane, nullptr, TCO(0));
}

// ---- one visit of register level LI (NC cells per side) and everything below it ---------------------------------------
// St: the per-level register state, a struct with members U0/F0 .. U3/F3 (T[4] each); LI picks the pair.
template <typename T> struct T2State { T U0[4], F0[4], U1[4], F1[4], U2[4], F2[4], U3[4], F3[4]; };
template <int LI, typename T> __device__ __forceinline__ T (&t2_U(T2State<T>& s))[4] {
  if constexpr (LI == 0) return s.U0; else if constexpr (LI == 1) return s.U1; else if constexpr (LI == 2) return s.U2; else return s.U3;
}
template <int LI, typename T> __device__ __forceinline__ T (&t2_F(T2State<T>& s))[4] {
  if constexpr (LI == 0) return s.F0; else if constexpr (LI == 1) return s.F1; else if constexpr (LI == 2) return s.F2; else return s.F3;
}

template <int LI, int NC, typename T, typename TCO, typename TC, int SM, int NCTOP>
__device__ __forceinline__ void t2_visit(T2Ctx<T, TCO, TC, SM, NCTOP>& cx, T2State<T>& st, bool zero) {
  using G = T2Geo<NC>;
  using L = T2Lds<T, TCO, NCTOP>;
  constexpr bool LAST = (NC == 8);                  // the level below is the 5 x 5 coarsest level (dtype TCO)
  using TX = typename std::conditional<LAST, TCO, T>::type;
  const Tail2Args& a = cx.a;
  const T2Lane<T> p = t2_lane<T, NC>();
  T (&U)[4] = t2_U<LI>(st);
  T (&F)[4] = t2_F<LI>(st);
  const Tail2Level& lv = a.lv[LI];
  T2Const<T> c;
  c.ihx2 = (T)lv.ihx2; c.ihy2 = (T)lv.ihy2; c.invD = (T)lv.invD; c.D = (T)lv.diag;
  c.omega = (T)a.omega; c.one_m_omega = (T)(1.0 - a.omega); c.coeff = (T)a.coeff; c.use_div = lv.use_div != 0; c.coff = a.colour_offset;
  if (zero) {
#pragma unroll
    for (int k = 0; k < 4; ++k) U[k] = T(0);
  }
  // down leg: pre sweeps -> residual -> full weighting into the level below
  t2_smooth<T, NC, SM>(U, F, cx.xbuf, cx.stage, p, c, a.pre);
  TX* fc = LAST ? reinterpret_cast<TX*>(cx.pool + L::kFive) + 25 : reinterpret_cast<TX*>(cx.pool + L::off_f(NC / 2));
  TX* ec = LAST ? reinterpret_cast<TX*>(cx.pool + L::kFive) : reinterpret_cast<TX*>(cx.pool + L::off_e(NC / 2));
  t2_residual_restrict<T, TX, NC>(U, F, cx.xbuf, cx.stage, p, c, fc);
  t2_sync<G::BLOCK>();
  // the level(s) below
  if constexpr (LAST) {
    if (!G::BLOCK || threadIdx.x < 64) {
      for (int r = 0; r < a.reps[LI]; ++r) t2_solve5<T, TCO, TC, SM, NCTOP>(cx, r == 0);
    }
  } else {
    constexpr int NCC = NC / 2;
    using GC = T2Geo<NCC>;
    auto below = [&]() {
      // the coarse level's lanes take their right-hand side from the staging array
      const T2Lane<T> q = t2_lane<T, NCC>();
      T (&Fc)[4] = t2_F<LI + 1>(st);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = 4 * q.strip + k;
        Fc[k] = (q.colok && i >= 1) ? reinterpret_cast<const T*>(fc)[i * (NCC + 1) + q.col] : T(0);
      }
      for (int r = 0; r < a.reps[LI]; ++r) t2_visit<LI + 1, NCC, T, TCO, TC, SM, NCTOP>(cx, st, r == 0);
      // ... and publish their iterate for the interpolation
      T (&Uc)[4] = t2_U<LI + 1>(st);
      if (q.active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<T*>(ec)[(4 * q.strip + k) * (NCC + 1) + q.col] = Uc[k];
      }
    };
    if (G::BLOCK && !GC::BLOCK) {          // the levels below are one wave's: the other waves go straight to the barrier
      if (threadIdx.x < 64) below();
    } else {
      below();
    }
  }
  t2_sync<G::BLOCK>();
  // up leg: u += P e -> post sweeps
  t2_prolong_add<T, TX, TC, NC>(U, p, ec);
  t2_smooth<T, NC, SM>(U, F, cx.xbuf, cx.stage, p, c, a.post);
}

template <typename T, typename TCO, typename TC, int SM, int NCTOP>
__global__ __launch_bounds__(T2Geo<NCTOP>::WAVES * 64) void tail2_kernel(const T* __restrict__ rhs_top, T* __restrict__ u_top,
                                                                          Tail2Args a, int zero_top, int* __restrict__ sweeps_out) {
  using G = T2Geo<NCTOP>;
  using L = T2Lds<T, TCO, NCTOP>;
  extern __shared__ __attribute__((aligned(16))) unsigned char pool[];
  constexpr int LI0 = 0;
  T2Ctx<T, TCO, TC, SM, NCTOP> cx{a, pool, reinterpret_cast<T*>(pool + L::kX), 0, 0, {0, 0, 0, 0, 0, 0, 0, 0, 0}};
  {   // zero the staging arrays (their far rows / columns and rings stay zero for the whole launch)
    int4* w = reinterpret_cast<int4*>(pool);
    for (int q = threadIdx.x; q < (int)(L::kTotal / 16); q += G::WAVES * 64) w[q] = make_int4(0, 0, 0, 0);
  }
  __syncthreads();
  const T2Lane<T> p = t2_lane<T, NCTOP>();
  T2State<T> st;
  T (&U)[4] = t2_U<LI0>(st);
  T (&F)[4] = t2_F<LI0>(st);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * p.strip + k;
    const bool ok = p.colok && i >= 1;
    F[k] = ok ? rhs_top[(size_t)i * a.ld_top + p.col] : T(0);
    U[k] = (ok && !zero_top) ? u_top[(size_t)i * a.ld_top + p.col] : T(0);
  }
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (a.direct) {
      const int li = lane < 9 ? lane : 0;
#pragma unroll
      for (int j = 0; j < 9; ++j) cx.minv_row[j] = a.minv[li * 9 + j];
    } else if (lane < 25) {
      // the ring of the coarsest right-hand side (injected ring of f): the stop test of the iteration counts it
      const int i = lane / 5, j = lane - 5 * i;
      TCO* sf = reinterpret_cast<TCO*>(pool + L::kFive) + 25;
      if (i == 0 || i == 4 || j == 0 || j == 4) sf[lane] = reinterpret_cast<const TCO*>(a.ring5)[(size_t)i * a.ring5_ld + j];
    }
  }
  __syncthreads();
  t2_visit<LI0, NCTOP, T, TCO, TC, SM, NCTOP>(cx, st, zero_top != 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * p.strip + k;
    if (p.colok && i >= 1) u_top[(size_t)i * a.ld_top + p.col] = U[k];
  }
  if (threadIdx.x == 0 && sweeps_out) *sweeps_out = cx.sweeps;
}

}  // namespace mg
