// Register-resident coarse tail: every level of at most 65^2 cells, down to and including the 5 x 5 coarsest solve, in ONE
// workgroup per visit -- the sub-cycle a W- or F-cycle enters 2^l times per fine-grid cycle.
//
// coarse_tail_kernel (mg_kernels.hpp) keeps these levels in LDS and pays ~0.4 us per barrier-separated stage: every cell
// update is five LDS reads behind index arithmetic that runs on run-time level geometry.  Here the levels are the squares
// 2^m + 1 every dyadic hierarchy ends in -- 65, 33, 17, 9, 5 -- with COMPILE-TIME geometry, and the iterate and the
// right-hand side of every level live in REGISTERS for the whole visit:
//
//   * a level of n = NC + 1 points per side is the NC x NC cells (i, j), 0 <= i, j < NC: row 0 and column 0 are boundary
//     cells (they hold 0 and are never updated), row / column NC is the far boundary (never stored).  A lane owns R
//     consecutive rows ("strip" s = rows R s .. R s + R - 1; R = 4 on the 65^2 level, 2 below) of TWO adjacent columns
//     2 c, 2 c + 1; a wave holds 128 / NC strips side by side (lane = sub-strip * NC / 2 + c): 65^2 is 8 waves, 33^2 four,
//     17^2 one, 9^2 sixteen lanes;
//   * lateral neighbours: the lane's other column, or lane -+ 1 by DPP wave_shr:1 / wave_shl:1.  Sub-strips sit side by
//     side in a wave, so the lane right of a sub-strip's last column is the NEXT sub-strip's column 0 (or nothing:
//     bound_ctrl zero) -- a boundary cell, value 0, exactly what the stencil needs there;
//   * red-black GS: with two columns per lane the colour of a lane's cells is the same in every lane -- cell e of row k is
//     red iff (k + e + colour_offset) is even -- so a colour pass updates ONE cell per row, with one DPP move, and touches
//     no cell of the other colour: half the arithmetic of a masked full sweep;
//   * vertical neighbours: the lane's own rows; across strips the first / last row of every strip crosses a small LDS
//     exchange buffer once per stage (one 16-byte write pair, one sync, one read pair; double-buffered by stage parity);
//     on the 9^2 level the four strips share a 16-lane DPP row and the exchange is row_shr:4 / row_shl:4, no LDS at all;
//   * every stage is written layer by layer over the lane's rows (all sums, then all products, ...): ONE wave has nobody to
//     hide its instruction latency behind, so the rows' dependency chains have to interleave in the instruction stream;
//   * full weighting in the fine layout (rows k -+ 1 of the residual strip, DPP for the west column), coarse values staged
//     through LDS into the coarse level's lanes; bilinear interpolation from the coarse iterate staged in LDS (n x n with
//     its zero ring, so the far-edge cells read zeros);
//   * levels of one wave (17^2, 9^2, 5^2) synchronise with wave-level fences only: the other waves wait at the next
//     workgroup barrier of the level above;
//   * the 5 x 5 solve is lexgs_5x5_zero_ring (mg_kernels.hpp: the reference's lexicographic Gauss-Seidel to coarse_tol,
//     bit for bit) or, mg_config.coarse_direct, u = A^-1 f;
//   * variable coefficients (VAR, from 33^2 down): the face means of a lane's cells and their reciprocal diagonal are formed
//     once per launch from the levels' coefficient / rdiag fields and live in registers beside the iterate (T2Coef).
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
  // variable coefficients (VAR): vertex values of a and the reciprocal diagonal of every register level (dtype T, pitch
  // a_ld[i]), and the 5 x 5 level's coefficient (dtype TCO)
  const void* a_lv[kT2MaxLev + 1];
  const void* rd_lv[kT2MaxLev];
  int a_ld[kT2MaxLev + 1];
};

template <int NC> struct T2Geo {
  static constexpr int N = NC + 1;              // points per side
  static constexpr int R = (NC >= 64) ? 4 : 2;  // rows per lane
  static constexpr int LP = NC / 2;             // lanes per strip row (two columns each)
  static constexpr int SPW = 64 / LP;           // strips per wave
  static constexpr int NS = NC / R;             // strips of R rows
  static constexpr int WAVES = (NS + SPW - 1) / SPW;
  static constexpr bool BLOCK = WAVES > 1;      // more than one wave: workgroup barriers; else wave-level fences
  static constexpr int XELEMS = NS * 2 * NC;    // one exchange buffer (elements)
};

// timing experiment (-DMG_EXPERIMENTS -DMG_EXP_TAIL_TRACE=1): wall-clock stamps of the last launch, (id, ticks) pairs;
// id = 10 * register level + phase (0 enter, 1 pre sweeps done, 2 residual + restriction done, 3 levels below done,
// 4 interpolation done, 5 post sweeps done), 90 / 91 / 92: kernel entry / top level loaded / stored
#if MG_EXP_TAIL_TRACE
__device__ long long g_tail2_trace[2 * 1024 + 2];
#define T2_STAMP(id)                                                                          \
  do {                                                                                        \
    if (threadIdx.x == 0 && cx.nstamp < 1024) {                                               \
      g_tail2_trace[2 + 2 * cx.nstamp] = (id);                                                \
      g_tail2_trace[3 + 2 * cx.nstamp] = wall_clock64();                                      \
      g_tail2_trace[0] = ++cx.nstamp;                                                         \
    }                                                                                         \
  } while (0)
#else
#define T2_STAMP(id) do { } while (0)
#endif

struct T2Lane {                                  // where this lane sits on a level
  int lc, strip;                                // column pair (columns 2 lc, 2 lc + 1), strip
  bool active;                                  // the lane holds cells of this level
  bool ok0;                                     // its first column is not the boundary column (lc >= 1)
};
template <int NC> __device__ __forceinline__ T2Lane t2_lane() {
  using G = T2Geo<NC>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T2Lane p;
  p.lc = lane & (G::LP - 1);
  p.strip = wave * G::SPW + lane / G::LP;
  p.active = p.strip < G::NS;
  p.ok0 = p.active && p.lc >= 1;
  return p;
}

template <bool BLOCK> __device__ __forceinline__ void t2_sync() {
  if (BLOCK) __syncthreads();
  else wave_lds_fence<int>();
}

template <typename T> struct alignas(2 * sizeof(T)) T2Pair { T v[2]; };

// first / last row of every strip to the strips above / below: `above` = last row of the strip above, `below` = first row of
// the strip below (zeros beyond the level)
template <typename T, int NC>
__device__ __forceinline__ void t2_exchange(T* __restrict__ xb, const T2Lane& p, const T (&top)[2], const T (&bottom)[2], T (&above)[2],
                                            T (&below)[2]) {
  using G = T2Geo<NC>;
  if (NC == 8) {
    // 9^2: the four strips sit in lanes 0..3, 4..7, 8..11, 12..15 of ONE 16-lane DPP row: the neighbour strip is a row shift
    // by 4 away (bound_ctrl zeros beyond the row = the boundary rows), no LDS round trip and no fence in the stage
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      above[e] = dpp_row_move<0x114>(bottom[e]);          // row_shr:4: lane l receives lane l - 4
      below[e] = dpp_row_move<0x104>(top[e]);             // row_shl:4: lane l receives lane l + 4
    }
    return;
  }
  using P = T2Pair<T>;
  P* x = reinterpret_cast<P*>(xb);
  if (p.active) {
    P t, b;
    t.v[0] = top[0]; t.v[1] = top[1]; b.v[0] = bottom[0]; b.v[1] = bottom[1];
    x[(p.strip * 2 + 0) * G::LP + p.lc] = t;
    x[(p.strip * 2 + 1) * G::LP + p.lc] = b;
  }
  t2_sync<G::BLOCK>();
  P a, bl;
  a.v[0] = a.v[1] = bl.v[0] = bl.v[1] = T(0);
  if (p.active && p.strip > 0) a = x[((p.strip - 1) * 2 + 1) * G::LP + p.lc];
  if (p.active && p.strip < G::NS - 1) bl = x[((p.strip + 1) * 2 + 0) * G::LP + p.lc];
  above[0] = a.v[0]; above[1] = a.v[1]; below[0] = bl.v[0]; below[1] = bl.v[1];
}

// Variable coefficients: the face means of the lane's cells -- av[k][e]: between rows k - 1 and k of column e (a(i-1/2) of
// row k, a(i+1/2) of row k - 1), ah[k][q]: west of column 0 / between the two columns / east of column 1 -- and their
// reciprocal diagonal (the level's rdiag field), formed once per launch and kept in registers for every visit of the level.
template <typename T, int R> struct T2Coef { T av[R + 1][2], ah[R][3], rd[R][2]; };
template <typename T> struct T2NoCoef {};

template <typename T, int NC>
__device__ __forceinline__ void t2_load_coef(T2Coef<T, T2Geo<NC>::R>& cf, const T2Lane& p, const T* __restrict__ a, const T* __restrict__ rd,
                                             int ld) {
  constexpr int R = T2Geo<NC>::R;
  T A[R + 2][4];                                     // rows R s - 1 .. R s + R, columns 2 lc - 1 .. 2 lc + 2 (clamped at 0)
#pragma unroll
  for (int r = 0; r < R + 2; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = max(R * p.strip - 1 + r, 0), j = max(2 * p.lc - 1 + q, 0);
      A[r][q] = p.active ? a[(size_t)i * ld + j] : T(0);
    }
#pragma unroll
  for (int k = 0; k <= R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) cf.av[k][e] = T(0.5) * (A[k + 1][e + 1] + A[k][e + 1]);
#pragma unroll
  for (int k = 0; k < R; ++k) {
#pragma unroll
    for (int q = 0; q < 3; ++q) cf.ah[k][q] = T(0.5) * (A[k + 1][q + 1] + A[k + 1][q]);
#pragma unroll
    for (int e = 0; e < 2; ++e) cf.rd[k][e] = p.active ? rd[(size_t)(R * p.strip + k) * ld + 2 * p.lc + e] : T(0);
  }
}

// Per-level constants.  They are wave-uniform, but kept in VECTOR registers on purpose (t2_pin): as kernel arguments the
// compiler re-loads them with s_load inside the sweep loops whenever scalar registers run short -- a ~200-cycle round trip
// per stage of a kernel whose stages are a few hundred cycles long.
template <typename T> struct T2Const { T ihx2, ihy2, invD, D, omega, one_m_omega, coeff; int coff; };
template <typename T> __device__ __forceinline__ T t2_pin(T x) {
  asm volatile("" : "+v"(x));
  return x;
}

// One weighted-Jacobi sweep.  DIV: divide by the diagonal (the reference's expression; needed when 1 / D is not exact), else
// multiply by the exact 1 / D.  Written layer by layer over the 2 R cells of the lane.
template <typename T, int NC, bool DIV, bool VAR, typename CF>
__device__ __forceinline__ void t2_jacobi(T (&U)[T2Geo<NC>::R][2], const T (&F)[T2Geo<NC>::R][2], T* __restrict__ xbuf, int& stage,
                                          const T2Lane& p, const T2Const<T>& c, const CF& cf) {
  using G = T2Geo<NC>;
  constexpr int R = G::R;
  T above[2], below[2];
  t2_exchange<T, NC>(xbuf + (stage & 1) * G::XELEMS, p, U[0], U[R - 1], above, below);
  ++stage;
  T sx[R][2], sy[R][2], t1[R][2];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const T west = dpp_from_lower_lane<T>(U[k][1]);             // every lane active here
    const T east = dpp_from_upper_lane<T>(U[k][0]);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const T up = (k > 0) ? U[k > 0 ? k - 1 : 0][e] : above[e];
      const T dn = (k < R - 1) ? U[k < R - 1 ? k + 1 : 0][e] : below[e];
      if constexpr (VAR) sx[k][e] = cf.av[k + 1][e] * dn + cf.av[k][e] * up;          // aip dn + aim up
      else sx[k][e] = dn + up;
    }
    if constexpr (VAR) {
      sy[k][0] = cf.ah[k][1] * U[k][1] + cf.ah[k][0] * west;    // ajp east + ajm west
      sy[k][1] = cf.ah[k][2] * east + cf.ah[k][1] * U[k][0];
    } else {
      sy[k][0] = U[k][1] + west;                                // (east + west)
      sy[k][1] = east + U[k][0];
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) sx[k][e] = c.ihx2 * sx[k][e];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) sy[k][e] = c.ihy2 * sy[k][e];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) sx[k][e] = sx[k][e] + sy[k][e];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) sx[k][e] = F[k][e] + sx[k][e];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if constexpr (VAR) sx[k][e] = sx[k][e] * cf.rd[k][e];
      else sx[k][e] = DIV ? sx[k][e] / c.D : sx[k][e] * c.invD;
    }
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) t1[k][e] = c.one_m_omega * U[k][e];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) sx[k][e] = c.omega * sx[k][e];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int e = 0; e < 2; ++e) sx[k][e] = t1[k][e] + sx[k][e];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const bool rowok = (k > 0 || p.strip > 0);
    U[k][0] = (p.ok0 && rowok) ? sx[k][0] : U[k][0];
    U[k][1] = (p.active && rowok) ? sx[k][1] : U[k][1];
  }
}

// One colour pass of red-black GS (colour 0 = (i + j + colour_offset) even first), in place.  Q = (colour + colour_offset) & 1:
// row k updates its cell e = (Q + k) & 1 -- the same e in every lane -- from cells of the other colour only.
template <typename T, int NC, bool DIV, int Q, bool VAR, typename CF>
__device__ __forceinline__ void t2_rb_pass(T (&U)[T2Geo<NC>::R][2], const T (&F)[T2Geo<NC>::R][2], T* __restrict__ xbuf, int& stage,
                                           const T2Lane& p, const T2Const<T>& c, const CF& cf) {
  using G = T2Geo<NC>;
  constexpr int R = G::R;
  T above[2], below[2];
  t2_exchange<T, NC>(xbuf + (stage & 1) * G::XELEMS, p, U[0], U[R - 1], above, below);
  ++stage;
  T sx[R], sy[R], t1[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    constexpr int e0 = Q & 1;
    const int e = (e0 + k) & 1;                       // folds per unrolled k
    const T up = (k > 0) ? U[k > 0 ? k - 1 : 0][e] : above[e];
    const T dn = (k < R - 1) ? U[k < R - 1 ? k + 1 : 0][e] : below[e];
    // (east + west): one of the two is the lane's other column, the other one comes from the neighbour lane
    if constexpr (VAR) {
      sx[k] = cf.av[k + 1][e] * dn + cf.av[k][e] * up;
      if (e == 0) sy[k] = cf.ah[k][1] * U[k][1] + cf.ah[k][0] * dpp_from_lower_lane<T>(U[k][1]);
      else sy[k] = cf.ah[k][2] * dpp_from_upper_lane<T>(U[k][0]) + cf.ah[k][1] * U[k][0];
    } else {
      sx[k] = dn + up;
      if (e == 0) sy[k] = U[k][1] + dpp_from_lower_lane<T>(U[k][1]);
      else sy[k] = dpp_from_upper_lane<T>(U[k][0]) + U[k][0];
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) sx[k] = c.ihx2 * sx[k];
#pragma unroll
  for (int k = 0; k < R; ++k) sy[k] = c.ihy2 * sy[k];
#pragma unroll
  for (int k = 0; k < R; ++k) sx[k] = sx[k] + sy[k];
#pragma unroll
  for (int k = 0; k < R; ++k) sx[k] = F[k][(Q + k) & 1] + sx[k];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if constexpr (VAR) sx[k] = sx[k] * cf.rd[k][(Q + k) & 1];
    else sx[k] = DIV ? sx[k] / c.D : sx[k] * c.invD;
  }
#pragma unroll
  for (int k = 0; k < R; ++k) t1[k] = c.one_m_omega * U[k][(Q + k) & 1];
#pragma unroll
  for (int k = 0; k < R; ++k) sx[k] = c.omega * sx[k];
#pragma unroll
  for (int k = 0; k < R; ++k) sx[k] = t1[k] + sx[k];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int e = (Q + k) & 1;
    const bool ok = (e == 0 ? p.ok0 : p.active) && (k > 0 || p.strip > 0);
    U[k][e] = ok ? sx[k] : U[k][e];
  }
}

template <typename T, int NC, int SM, bool DIV, bool VAR, typename CF>
__device__ __forceinline__ void t2_smooth(T (&U)[T2Geo<NC>::R][2], const T (&F)[T2Geo<NC>::R][2], T* __restrict__ xbuf, int& stage,
                                          const T2Lane& p, const T2Const<T>& c, const CF& cf, int nsweep) {
  if (SM == kSmRbgs) {
    if (c.coff & 1) {
#pragma unroll 1
      for (int s = 0; s < nsweep; ++s) {
        t2_rb_pass<T, NC, DIV, 1, VAR>(U, F, xbuf, stage, p, c, cf);
        t2_rb_pass<T, NC, DIV, 0, VAR>(U, F, xbuf, stage, p, c, cf);
      }
    } else {
#pragma unroll 1
      for (int s = 0; s < nsweep; ++s) {
        t2_rb_pass<T, NC, DIV, 0, VAR>(U, F, xbuf, stage, p, c, cf);
        t2_rb_pass<T, NC, DIV, 1, VAR>(U, F, xbuf, stage, p, c, cf);
      }
    }
  } else {
#pragma unroll 1
    for (int s = 0; s < nsweep; ++s) t2_jacobi<T, NC, DIV, VAR>(U, F, xbuf, stage, p, c, cf);
  }
}

// residual of the strip, then full weighting of the interior coarse cells on it; the coarse values go to fc (the coarse
// level's n x n staging array, dtype TX) -- the caller synchronises before the coarse lanes read them
template <typename T, typename TX, int NC, bool VAR, typename CF>
__device__ __forceinline__ void t2_residual_restrict(const T (&U)[T2Geo<NC>::R][2], const T (&F)[T2Geo<NC>::R][2], T* __restrict__ xbuf,
                                                     int& stage, const T2Lane& p, const T2Const<T>& c, const CF& cf, T sigma,
                                                     TX* __restrict__ fc) {
  using G = T2Geo<NC>;
  constexpr int R = G::R;
  constexpr int NP = NC / 2 + 1;                  // coarse points per side
  T Rr[R][2];
  {
    T above[2], below[2];
    t2_exchange<T, NC>(xbuf + (stage & 1) * G::XELEMS, p, U[0], U[R - 1], above, below);
    ++stage;
    T sx[R][2], sy[R][2], md[R][2];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const T west = dpp_from_lower_lane<T>(U[k][1]);
      const T east = dpp_from_upper_lane<T>(U[k][0]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const T up = (k > 0) ? U[k > 0 ? k - 1 : 0][e] : above[e];
        const T dn = (k < R - 1) ? U[k < R - 1 ? k + 1 : 0][e] : below[e];
        if constexpr (VAR) sx[k][e] = cf.av[k + 1][e] * dn + cf.av[k][e] * up;
        else sx[k][e] = dn + up;
      }
      if constexpr (VAR) {
        sy[k][0] = cf.ah[k][1] * U[k][1] + cf.ah[k][0] * west;
        sy[k][1] = cf.ah[k][2] * east + cf.ah[k][1] * U[k][0];
      } else {
        sy[k][0] = U[k][1] + west;                                // (east + west)
        sy[k][1] = east + U[k][0];
      }
    }
    // r = f - coeff (((dn + up) ihx2 + (east + west) ihy2) - u D), layer by layer (VAR: sums weighted by the face means, D per cell)
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) sx[k][e] = sx[k][e] * c.ihx2;
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) sy[k][e] = sy[k][e] * c.ihy2;
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if constexpr (VAR) {
          const T D0 = (cf.av[k + 1][e] + cf.av[k][e]) * c.ihx2 + (cf.ah[k][e + 1] + cf.ah[k][e]) * c.ihy2;
          md[k][e] = U[k][e] * ((sigma != T(0)) ? D0 + sigma : D0);
        } else {
          md[k][e] = U[k][e] * c.D;
        }
      }
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) sx[k][e] = sx[k][e] + sy[k][e];
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) sx[k][e] = sx[k][e] - md[k][e];
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) sx[k][e] = c.coeff * sx[k][e];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const bool rowok = (k > 0 || p.strip > 0);
      Rr[k][0] = (p.ok0 && rowok) ? (F[k][0] - sx[k][0]) : T(0);        // boundary cells: no interior coarse cell reads them
      Rr[k][1] = (p.active && rowok) ? (F[k][1] - sx[k][1]) : T(0);
    }
  }
  {
    T above[2], below[2];
    t2_exchange<T, NC>(xbuf + (stage & 1) * G::XELEMS, p, Rr[0], Rr[R - 1], above, below);
    ++stage;
    (void)below;
#pragma unroll
    for (int k = 0; k < R; k += 2) {                // the even (coarse) rows of the strip; the coarse column is the lane's first
      const T n = (k > 0) ? Rr[k > 0 ? k - 1 : 0][0] : above[0], ne = (k > 0) ? Rr[k > 0 ? k - 1 : 0][1] : above[1];
      const T cc = Rr[k][0], ee = Rr[k][1];
      const T s = Rr[k + 1][0], se = Rr[k + 1][1];
      const T nw = dpp_from_lower_lane<T>(ne), w = dpp_from_lower_lane<T>(ee), sw = dpp_from_lower_lane<T>(se);
      const T corners = ((nw + ne) + sw) + se;
      const T edges = ((n + s) + w) + ee;
      const T val = (T(1.0 / 16.0) * corners + T(1.0 / 8.0) * edges) + T(1.0 / 4.0) * cc;
      const int i = R * p.strip + k;
      if (p.ok0 && i >= 2) fc[(i >> 1) * NP + p.lc] = (TX)val;
    }
  }
}

// u += P e: `ec` is the coarse iterate as an n x n array (zero ring) of dtype TX; interpolation in TC (the fine GRID's
// dtype), the sum in the wider of (T, TC), rounded to T (operators/transfer.py:234-267 + solvers/multigrid.py:329)
template <typename T, typename TX, typename TC, int NC>
__device__ __forceinline__ void t2_prolong_add(T (&U)[T2Geo<NC>::R][2], const T2Lane& p, const TX* __restrict__ ec) {
  using TS = typename std::conditional<(sizeof(TC) > sizeof(T)), TC, T>::type;
  constexpr int R = T2Geo<NC>::R;
  constexpr int NP = NC / 2 + 1;
  if (!p.active) return;
  const int ic0 = (R / 2) * p.strip;
  TC e0[R / 2 + 1], e1[R / 2 + 1];                   // coarse rows under the strip at coarse columns lc / lc + 1
#pragma unroll
  for (int r = 0; r <= R / 2; ++r) {
    e0[r] = (TC)ec[(ic0 + r) * NP + p.lc];
    e1[r] = (TC)ec[(ic0 + r) * NP + p.lc + 1];
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int r = k >> 1;
    TC v0, v1;                                        // even column 2 lc sits on coarse column lc; odd column between lc and lc + 1
    if ((k & 1) == 0) { v0 = e0[r]; v1 = TC(0.5) * (e0[r] + e1[r]); }
    else { v0 = TC(0.5) * (e0[r] + e0[r + 1]); v1 = TC(0.25) * (((e0[r] + e1[r]) + e0[r + 1]) + e1[r + 1]); }
    const bool rowok = (k > 0 || p.strip > 0);
    if (p.ok0 && rowok) U[k][0] = (T)((TS)U[k][0] + (TS)v0);
    if (rowok) U[k][1] = (T)((TS)U[k][1] + (TS)v1);
  }
}

// LDS layout of one launch (byte offsets into the dynamic pool), for the top level NCTOP
template <typename T, typename TCO, int NCTOP> struct T2Lds {
  static constexpr size_t al(size_t x) { return (x + 15) / 16 * 16; }
  static constexpr size_t kX = 0;                                                     // 2 exchange buffers of the top level
  static constexpr size_t kXBytes = al(2 * (size_t)T2Geo<NCTOP>::XELEMS * sizeof(T));
  // staging arrays of the coarser levels: a level of NC cells per side has an (NC + 1)^2 array for its rhs (written by the
  // level above) and one for its iterate (read by the level above)
  static constexpr size_t lev_bytes(int nc) { return al((size_t)(nc + 1) * (nc + 1) * sizeof(T)); }
  static constexpr size_t off_f(int nc) {          // nc = NCTOP / 2, / 4, ... down to 8
    size_t o = kX + kXBytes;
    for (int m = NCTOP / 2; m > nc; m /= 2) o += 2 * lev_bytes(m);
    return o;
  }
  static constexpr size_t off_e(int nc) { return off_f(nc) + lev_bytes(nc); }
  static constexpr size_t kFive = off_f(4);                                           // su[25], sf[25], sa[25] (VAR) in TCO
  static constexpr size_t kTotal = kFive + al(3 * 25 * sizeof(TCO));
};

template <typename T, typename TCO, typename TC, int SM, int NCTOP, bool DIV, bool VAR> struct T2Ctx {
  const Tail2Args& a;
  unsigned char* pool;
  T* xbuf;
  int stage;
  int sweeps;
  T omega, one_m_omega, coeff;   // pinned in vector registers (t2_pin)
  int pre, post, coff, direct;
  int nstamp;                    // timing experiment (T2_STAMP)
  double minv_row[9];            // direct: row `lane` of the 9 x 9 inverse (lanes 0..8 of wave 0)
  T lvc[kT2MaxLev][4];           // ihx2, ihy2, 1 / D, D of every register level, pinned once at kernel entry (a visit that
                                 // fetched them from the kernel arguments paid an s_load round trip, ~0.3 us, every time)
  int reps[kT2MaxLev];
  T sigma;
};

// ---- the 5 x 5 level: rhs = full weighting of the 9^2 level's residual (already in sf), solve, result in su -----------
template <typename CX, typename T, typename TCO, int NCTOP, bool VAR>
__device__ __forceinline__ void t2_solve5(CX& cx, bool zero) {
  using L = T2Lds<T, TCO, NCTOP>;
  TCO* su = reinterpret_cast<TCO*>(cx.pool + L::kFive);
  TCO* sf = su + 25;
  const int lane = threadIdx.x & 63;
  const Tail2Args& a = cx.a;
  if (cx.direct) {
    // nine unknowns in lanes 0..8: u_i = sum_j minv[i][j] f_j, f_j broadcast from lane j; a fixed summation order.  The
    // result does not depend on the iterate the level starts from (nor on how often it is visited with one right-hand side)
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
  if (zero) {
    if (lane < 25) su[lane] = TCO(0);
    wave_lds_fence<int>();
  }
  const TCO hx2 = (TCO)a.hx2_5, hy2 = (TCO)a.hy2_5, diag = (TCO)a.diag_5, cf = (TCO)a.coeff;
  const TCO* sa = VAR ? sf + 25 : nullptr;
  const TCO sg = VAR ? (TCO)a.sigma : TCO(0);
  cx.sweeps = a.exact_5
      ? lexgs_5x5_zero_ring<TCO, VAR, true>(su, sf, hx2, hy2, diag, cf, TCO(1), TCO(0), a.hxhy_5, a.tol_x, a.maxit, lane, sa, sg)
      : lexgs_5x5_zero_ring<TCO, VAR, false>(su, sf, hx2, hy2, diag, cf, TCO(1), TCO(0), a.hxhy_5, a.tol_x, a.maxit, lane, sa, sg);
}

// ---- the per-level register state: iterate and rhs of register level LI (R x 2 values each) ---------------------------
template <typename T, int NCTOP, bool VAR> struct T2State {
  T U0[T2Geo<NCTOP>::R][2], F0[T2Geo<NCTOP>::R][2];
  T U1[2][2], F1[2][2], U2[2][2], F2[2][2], U3[2][2], F3[2][2];     // the levels below the top have at most 32 cells per side: R = 2
  typename std::conditional<VAR, T2Coef<T, T2Geo<NCTOP>::R>, T2NoCoef<T>>::type C0;
  typename std::conditional<VAR, T2Coef<T, 2>, T2NoCoef<T>>::type C1, C2, C3;
};
template <int LI, typename S> __device__ __forceinline__ auto& t2_U(S& s) {
  if constexpr (LI == 0) return s.U0; else if constexpr (LI == 1) return s.U1; else if constexpr (LI == 2) return s.U2; else return s.U3;
}
template <int LI, typename S> __device__ __forceinline__ auto& t2_F(S& s) {
  if constexpr (LI == 0) return s.F0; else if constexpr (LI == 1) return s.F1; else if constexpr (LI == 2) return s.F2; else return s.F3;
}
template <int LI, typename S> __device__ __forceinline__ auto& t2_C(S& s) {
  if constexpr (LI == 0) return s.C0; else if constexpr (LI == 1) return s.C1; else if constexpr (LI == 2) return s.C2; else return s.C3;
}

// ---- one visit of register level LI (NC cells per side) and everything below it ---------------------------------------
template <int LI, int NC, typename T, typename TCO, typename TC, int SM, int NCTOP, bool DIV, bool VAR>
__device__ __forceinline__ void t2_visit(T2Ctx<T, TCO, TC, SM, NCTOP, DIV, VAR>& cx, T2State<T, NCTOP, VAR>& st, bool zero) {
  using G = T2Geo<NC>;
  using L = T2Lds<T, TCO, NCTOP>;
  constexpr int R = G::R;
  constexpr bool LAST = (NC == 8);                  // the level below is the 5 x 5 coarsest level (dtype TCO)
  using TX = typename std::conditional<LAST, TCO, T>::type;
  const Tail2Args& a = cx.a;
  const T2Lane p = t2_lane<NC>();
  auto& U = t2_U<LI>(st);
  auto& F = t2_F<LI>(st);
  const auto& cf = t2_C<LI>(st);
  T2Const<T> c;
  c.ihx2 = cx.lvc[LI][0]; c.ihy2 = cx.lvc[LI][1]; c.invD = cx.lvc[LI][2]; c.D = cx.lvc[LI][3];
  c.omega = cx.omega; c.one_m_omega = cx.one_m_omega; c.coeff = cx.coeff; c.coff = cx.coff;
  if (zero) {
#pragma unroll
    for (int k = 0; k < R; ++k) U[k][0] = U[k][1] = T(0);
  }
  T2_STAMP(10 * LI + 0);
  // down leg: pre sweeps -> residual -> full weighting into the level below
  t2_smooth<T, NC, SM, DIV, VAR>(U, F, cx.xbuf, cx.stage, p, c, cf, cx.pre);
  T2_STAMP(10 * LI + 1);
  TX* fc = LAST ? reinterpret_cast<TX*>(cx.pool + L::kFive) + 25 : reinterpret_cast<TX*>(cx.pool + L::off_f(NC / 2));
  TX* ec = LAST ? reinterpret_cast<TX*>(cx.pool + L::kFive) : reinterpret_cast<TX*>(cx.pool + L::off_e(NC / 2));
  t2_residual_restrict<T, TX, NC, VAR>(U, F, cx.xbuf, cx.stage, p, c, cf, cx.sigma, fc);
  t2_sync<G::BLOCK>();
  T2_STAMP(10 * LI + 2);
  // the level(s) below
  if constexpr (LAST) {
    const int reps = cx.direct ? 1 : cx.reps[LI];     // a direct solve repeated on the same right-hand side returns the same bits
    for (int r = 0; r < reps; ++r) t2_solve5<T2Ctx<T, TCO, TC, SM, NCTOP, DIV, VAR>, T, TCO, NCTOP, VAR>(cx, r == 0);
  } else {
    constexpr int NCC = NC / 2;
    using GC = T2Geo<NCC>;
    auto below = [&]() {
      // the coarse level's lanes take their right-hand side from the staging array
      const T2Lane q = t2_lane<NCC>();
      auto& Fc = t2_F<LI + 1>(st);
#pragma unroll
      for (int k = 0; k < GC::R; ++k) {
        const int i = GC::R * q.strip + k;
        const T* row = reinterpret_cast<const T*>(fc) + i * (NCC + 1) + 2 * q.lc;
        Fc[k][0] = (q.ok0 && i >= 1) ? row[0] : T(0);
        Fc[k][1] = (q.active && i >= 1) ? row[1] : T(0);
      }
      const int reps = cx.reps[LI];
      for (int r = 0; r < reps; ++r) t2_visit<LI + 1, NCC, T, TCO, TC, SM, NCTOP, DIV, VAR>(cx, st, r == 0);
      // ... and publish their iterate for the interpolation
      auto& Uc = t2_U<LI + 1>(st);
      if (q.active) {
#pragma unroll
        for (int k = 0; k < GC::R; ++k) {
          T* row = reinterpret_cast<T*>(ec) + (GC::R * q.strip + k) * (NCC + 1) + 2 * q.lc;
          row[0] = Uc[k][0];
          row[1] = Uc[k][1];
        }
      }
    };
    if (G::BLOCK && !GC::BLOCK) {          // the levels below are one wave's: the other waves go straight to the barrier
      if (threadIdx.x < 64) below();
    } else {
      below();
    }
  }
  t2_sync<G::BLOCK>();
  T2_STAMP(10 * LI + 3);
  // up leg: u += P e -> post sweeps
  t2_prolong_add<T, TX, TC, NC>(U, p, ec);
  T2_STAMP(10 * LI + 4);
  t2_smooth<T, NC, SM, DIV, VAR>(U, F, cx.xbuf, cx.stage, p, c, cf, cx.post);
  T2_STAMP(10 * LI + 5);
}

template <typename T, int NC, int LI, typename S>
__device__ __forceinline__ void t2_load_coefs(S& st, const Tail2Args& a) {
  const T2Lane q = t2_lane<NC>();
  if (!T2Geo<NC>::BLOCK && threadIdx.x >= 64) return;     // a one-wave level
  t2_load_coef<T, NC>(t2_C<LI>(st), q, reinterpret_cast<const T*>(a.a_lv[LI]), reinterpret_cast<const T*>(a.rd_lv[LI]), a.a_ld[LI]);
  if constexpr (NC > 8) t2_load_coefs<T, NC / 2, LI + 1>(st, a);
}

template <typename T, typename TCO, typename TC, int SM, int NCTOP, bool DIV, bool VAR>
__global__ __launch_bounds__(T2Geo<NCTOP>::WAVES * 64) void tail2_kernel(const T* __restrict__ rhs_top, T* __restrict__ u_top,
                                                                          Tail2Args a, int zero_top, int* __restrict__ sweeps_out) {
  using G = T2Geo<NCTOP>;
  using L = T2Lds<T, TCO, NCTOP>;
  using P = T2Pair<T>;
  constexpr int R = G::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char pool[];
  T2Ctx<T, TCO, TC, SM, NCTOP, DIV, VAR> cx{a, pool, reinterpret_cast<T*>(pool + L::kX), 0, 0,
                                            t2_pin((T)a.omega), t2_pin((T)(1.0 - a.omega)), t2_pin((T)a.coeff),
                                            a.pre, a.post, a.colour_offset, a.direct, 0, {0, 0, 0, 0, 0, 0, 0, 0, 0}, {}, {},
                                            VAR ? t2_pin((T)a.sigma) : T(0)};
#pragma unroll
  for (int l = 0; l < kT2MaxLev; ++l) {
    cx.lvc[l][0] = t2_pin((T)a.lv[l].ihx2); cx.lvc[l][1] = t2_pin((T)a.lv[l].ihy2);
    cx.lvc[l][2] = t2_pin((T)a.lv[l].invD); cx.lvc[l][3] = t2_pin((T)a.lv[l].diag);
    int r = a.reps[l];
    asm volatile("" : "+v"(r));
    cx.reps[l] = r;
  }
  T2_STAMP(90);
  // the top level first: its loads are in flight while the staging arrays are zeroed
  const T2Lane p = t2_lane<NCTOP>();
  T2State<T, NCTOP, VAR> st;
  auto& U = t2_U<0>(st);
  auto& F = t2_F<0>(st);
  int ld = a.ld_top;
  asm volatile("" : "+v"(ld));                     // one load of the argument, not one per (predicated) row
  P f2[R], u2[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int i = R * p.strip + k;
    f2[k].v[0] = f2[k].v[1] = u2[k].v[0] = u2[k].v[1] = T(0);
    if (p.active && i >= 1) {
      f2[k] = *reinterpret_cast<const P*>(rhs_top + (size_t)i * ld + 2 * p.lc);
      if (!zero_top) u2[k] = *reinterpret_cast<const P*>(u_top + (size_t)i * ld + 2 * p.lc);
    }
  }
  {   // zero the staging arrays (their far rows / columns and rings stay zero for the whole launch)
    int4* w = reinterpret_cast<int4*>(pool);
    for (int q = threadIdx.x; q < (int)(L::kTotal / 16); q += G::WAVES * 64) w[q] = make_int4(0, 0, 0, 0);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int t = threadIdx.x;
    if (a.direct) {
      const int li = t < 9 ? t : 0;
#pragma unroll
      for (int j = 0; j < 9; ++j) cx.minv_row[j] = a.minv[li * 9 + j];
    } else if (t < 25) {
      // the ring of the coarsest right-hand side (injected ring of f): the stop test of the iteration counts it
      const int i = t / 5, j = t - 5 * i;
      TCO* sf = reinterpret_cast<TCO*>(pool + L::kFive) + 25;
      if (i == 0 || i == 4 || j == 0 || j == 4) sf[t] = reinterpret_cast<const TCO*>(a.ring5)[(size_t)i * a.ring5_ld + j];
    }
    if (VAR && !a.direct && t < 25) {
      TCO* sa = reinterpret_cast<TCO*>(pool + L::kFive) + 50;
      sa[t] = reinterpret_cast<const TCO*>(a.a_lv[kT2MaxLev])[(size_t)(t / 5) * a.a_ld[kT2MaxLev] + t % 5];
    }
  }
  if constexpr (VAR) t2_load_coefs<T, NCTOP, 0>(st, a);
#pragma unroll
  for (int k = 0; k < R; ++k) {
    F[k][0] = p.ok0 ? f2[k].v[0] : T(0);            // column 0 is a boundary cell: no interior cell reads its rhs
    F[k][1] = f2[k].v[1];
    U[k][0] = p.ok0 ? u2[k].v[0] : T(0);
    U[k][1] = u2[k].v[1];
  }
  __syncthreads();
  T2_STAMP(91);
  t2_visit<0, NCTOP, T, TCO, TC, SM, NCTOP, DIV, VAR>(cx, st, zero_top != 0);
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int i = R * p.strip + k;
    if (p.active && i >= 1) {
      if (p.lc >= 1) {
        P o;
        o.v[0] = U[k][0]; o.v[1] = U[k][1];
        *reinterpret_cast<P*>(u_top + (size_t)i * ld + 2 * p.lc) = o;
      } else {
        u_top[(size_t)i * ld + 1] = U[k][1];        // column 0 is the boundary: not ours to write
      }
    }
  }
  if (threadIdx.x == 0 && sweeps_out) *sweeps_out = cx.sweeps;
  T2_STAMP(92);
}

}  // namespace mg
