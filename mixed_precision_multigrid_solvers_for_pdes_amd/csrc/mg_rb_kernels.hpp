// Register-blocked fused legs (weighted Jacobi / red-black GS, constant coefficients) for the bandwidth-bound levels.
//
// Same legs as fused_jacobi_kernel (mg_kernels.hpp) -- down: sweeps -> residual -> full-weighting restriction; up:
// u += P e -> sweeps [-> sum r^2]; plain: sweeps -- and the same arithmetic per cell (bit-identical results), but the
// iterate never lives in LDS:
//
//   * a workgroup of W waves owns a region of RI = W * RPT rows x 1 KB; every WAVE owns RPT consecutive rows of it, a
//     lane the same 16-byte column slot of each: u and rhs of the lane's RPT x N cells stay in registers for the whole
//     leg, loaded up front (2 RPT 16-byte loads per lane in flight: 16 KB per wave -- the "whole tile at once" access
//     shape that streams at the rate of a copy, profiles/README.md);
//   * vertical neighbours: rows k -+ 1 of the lane's own strip; across strips the top / bottom row of every wave goes
//     through a small LDS exchange buffer once per stage (2 writes + 2 reads of 16 bytes per lane and stage instead of
//     RPT writes + RPT + 2 reads + 2 RPT scalar reads of the LDS-tiled kernel);
//   * lateral neighbours: lane -+ 1 by DPP wave_shr:1 / wave_shl:1 -- a wave spans the whole region width, so the only
//     lanes without a neighbour are region-edge lanes, whose cells are halo (stale by construction);
//   * halo: HALO rows above / below, HL lanes (4 x 16 bytes) left / right: the tile is 56 lanes = 896 bytes = seven
//     128-byte lines wide, so every tile row starts on a line boundary;
//   * full weighting in registers too (rows k -+ 1 of the residual strip, DPP for the west column): no LDS r-tile;
//   * the legs are instruction-bound once the data streams (rocprofv3 SQ counters, profiles/README.md): a workgroup
//     whose region lies strictly inside the grid -- all but the rim of tiles -- takes the INTERIOR body, which drops
//     every per-cell bounds / boundary / far-edge test and knows row and column parities at compile time.
//
// LDS: the exchange buffer (2 x W x 2 KB) and, for the up leg, the coarse patch under the region.
#pragma once

#include "mg_kernels.hpp"

namespace mg {

// Halo LANES per side (16 bytes each).  4: the tile is 56 lanes = 896 bytes = seven 128-byte lines, every tile row starts
// on a line boundary.  2 (where 2 N cells cover the halo): 60 lanes = 960 bytes, tile rows start on 64-byte boundaries.
#ifndef MG_RB_HL_MIN
#define MG_RB_HL_MIN 4
#endif
template <typename T, int HALO, int W, int RPT> struct RbShape {
  static constexpr int N = VecW<T>::N;
  static constexpr int HL = (MG_RB_HL_MIN <= 2 && 2 * N >= HALO) ? 2 : ((HALO + N - 1) / N > 4 ? (HALO + N - 1) / N : 4);   // 5 for the fp64 red-black spanning leg (halo 10)
  static constexpr int RI = W * RPT;                 // region rows
  static constexpr int RJ = 64 * N;                  // region cols (one wave = one region row segment of 1 KB)
  static constexpr int TI = RI - 2 * HALO;           // tile rows
  static constexpr int TJ = (64 - 2 * HL) * N;       // tile cols
  static_assert(TI > 0 && TI % 2 == 0, "tile height must be even (coarse rows sit on every other tile row)");
  static_assert(RPT % 2 == 0, "rows per wave must be even (row parities are compile-time constants per strip row)");
  static_assert(HL * N >= HALO, "lateral halo too narrow");
};

#ifndef MG_RB_NT_MODE
#define MG_RB_NT_MODE 3          // stores + loads of u (measured best at 4097^2 fp64, profiles/README.md)
#endif
template <typename T> __device__ __forceinline__ T dpp_from_lower_lane(T x);     // lane l receives lane l-1's value (0 for lane 0)
template <typename T> __device__ __forceinline__ T dpp_from_upper_lane(T x);     // lane l receives lane l+1's value (0 for lane 63)
template <> __device__ __forceinline__ float dpp_from_lower_lane<float>(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));      // wave_shr:1
}
template <> __device__ __forceinline__ float dpp_from_upper_lane<float>(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true));      // wave_shl:1
}
template <> __device__ __forceinline__ double dpp_from_lower_lane<double>(double x) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true));
}
template <> __device__ __forceinline__ double dpp_from_upper_lane<double>(double x) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true));
}

// The per-row arithmetic of the constant-coefficient stages.  Generic form: the scalar expressions of the single-operator
// kernels, cell by cell.  fp32 form: the SAME expressions on two-cell vectors, which hipcc lowers to v_pk_add_f32 /
// v_pk_mul_f32 (two IEEE operations per lane and instruction, each rounded exactly like its scalar twin; no contraction:
// -ffp-contract=off) -- the fp32 legs are instruction-bound, so halving their arithmetic instructions is time.
template <typename T, int N> struct RowMath {
  // res = (1 - w) mid + w (f + ihx2 (dn + up) + ihy2 (east + west)) / D          (solvers/smoothers.py:62-84)
  static __device__ __forceinline__ Pack<T> relax(const Pack<T>& mid, const Pack<T>& dn, const Pack<T>& up, T left, T right,
                                                  const Pack<T>& f, T ihx2, T ihy2, T invD, T D, T omega, T one_m_omega, bool use_div) {
    Pack<T> res;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const T wv = (e == 0) ? left : mid.v[e > 0 ? e - 1 : 0];
      const T ea = (e == N - 1) ? right : mid.v[e < N - 1 ? e + 1 : 0];
      const T nb = ihx2 * (dn.v[e] + up.v[e]) + ihy2 * (ea + wv);
      const T un = use_div ? (f.v[e] + nb) / D : (f.v[e] + nb) * invD;
      res.v[e] = one_m_omega * mid.v[e] + omega * un;
    }
    return res;
  }
  // r = f - coeff (((dn + up) ihx2 + (east + west) ihy2) - mid D)                 (operators/laplacian.py:73-77, 117-118)
  static __device__ __forceinline__ Pack<T> resid(const Pack<T>& mid, const Pack<T>& dn, const Pack<T>& up, T left, T right,
                                                  const Pack<T>& f, T ihx2, T ihy2, T D, T coeff) {
    Pack<T> r;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const T wv = (e == 0) ? left : mid.v[e > 0 ? e - 1 : 0];
      const T ea = (e == N - 1) ? right : mid.v[e < N - 1 ? e + 1 : 0];
      const T au = coeff * (((dn.v[e] + up.v[e]) * ihx2 + (ea + wv) * ihy2) - mid.v[e] * D);
      r.v[e] = f.v[e] - au;
    }
    return r;
  }
};
template <> struct RowMath<float, 4> {
  typedef float f2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ Pack<float> relax(const Pack<float>& mid, const Pack<float>& dn, const Pack<float>& up, float left,
                                                      float right, const Pack<float>& f, float ihx2, float ihy2, float invD, float D,
                                                      float omega, float one_m_omega, bool use_div) {
    const f2 m0 = {mid.v[0], mid.v[1]}, m1 = {mid.v[2], mid.v[3]};
    const f2 sx0 = f2{dn.v[0], dn.v[1]} + f2{up.v[0], up.v[1]}, sx1 = f2{dn.v[2], dn.v[3]} + f2{up.v[2], up.v[3]};
    const f2 sy0 = f2{mid.v[1], mid.v[2]} + f2{left, mid.v[0]}, sy1 = f2{mid.v[3], right} + f2{mid.v[1], mid.v[2]};   // (east + west)
    const f2 nb0 = ihx2 * sx0 + ihy2 * sy0, nb1 = ihx2 * sx1 + ihy2 * sy1;
    const f2 t0 = f2{f.v[0], f.v[1]} + nb0, t1 = f2{f.v[2], f.v[3]} + nb1;
    const f2 un0 = use_div ? t0 / D : t0 * invD, un1 = use_div ? t1 / D : t1 * invD;
    const f2 r0 = one_m_omega * m0 + omega * un0, r1 = one_m_omega * m1 + omega * un1;
    Pack<float> res;
    res.v[0] = r0.x; res.v[1] = r0.y; res.v[2] = r1.x; res.v[3] = r1.y;
    return res;
  }
  static __device__ __forceinline__ Pack<float> resid(const Pack<float>& mid, const Pack<float>& dn, const Pack<float>& up, float left,
                                                      float right, const Pack<float>& f, float ihx2, float ihy2, float D, float coeff) {
    const f2 m0 = {mid.v[0], mid.v[1]}, m1 = {mid.v[2], mid.v[3]};
    const f2 sx0 = f2{dn.v[0], dn.v[1]} + f2{up.v[0], up.v[1]}, sx1 = f2{dn.v[2], dn.v[3]} + f2{up.v[2], up.v[3]};
    const f2 sy0 = f2{mid.v[1], mid.v[2]} + f2{left, mid.v[0]}, sy1 = f2{mid.v[3], right} + f2{mid.v[1], mid.v[2]};
    const f2 au0 = coeff * ((sx0 * ihx2 + sy0 * ihy2) - m0 * D), au1 = coeff * ((sx1 * ihx2 + sy1 * ihy2) - m1 * D);
    const f2 r0 = f2{f.v[0], f.v[1]} - au0, r1 = f2{f.v[2], f.v[3]} - au1;
    Pack<float> r;
    r.v[0] = r0.x; r.v[1] = r0.y; r.v[2] = r1.x; r.v[3] = r1.y;
    return r;
  }
};

// Exchange of the strips' edge rows: every wave publishes its first and last row, then reads the last row of the wave
// above and the first row of the wave below (zeros beyond the region).  `xb`: W x 2 x 64 packs.
template <typename T, int W>
__device__ __forceinline__ void rb_exchange(Pack<T>* __restrict__ xb, int w, int lane, const Pack<T>& top, const Pack<T>& bottom,
                                            Pack<T>& above, Pack<T>& below) {
  xb[(w * 2 + 0) * 64 + lane] = top;
  xb[(w * 2 + 1) * 64 + lane] = bottom;
  __syncthreads();
  above = (w > 0) ? xb[((w - 1) * 2 + 1) * 64 + lane] : zero_pack<T>();
  below = (w < W - 1) ? xb[((w + 1) * 2 + 0) * 64 + lane] : zero_pack<T>();
}

// INT: the region lies strictly inside the grid (and the coarse patch / restriction targets inside the coarse grid, the
// tile inside the norm window): no per-cell guard survives; only the region's own edge rows are skipped.
// NT (compile time -- behind a run-time flag the compiler merges the two stores and drops the hint): bit 0 non-temporal
// stores of the output tile, bit 1 non-temporal loads of u, bit 2 of rhs.  For arrays that cannot stay in the 256 MiB
// Infinity Cache from one leg to the next (4097^2 fp64: u, t and rhs are 3 x 136 MB) the hints keep the streamed
// operands from evicting each other; arrays that do fit (4097^2 fp32) are faster without them.
// VAR: the variable-coefficient operator (varcoef_kernel's discretisation and association order): the strip of `a` is
// loaded with u and rhs, its edge rows go through the exchange once, and the face means of the lane's cells --
// (RPT + 1) x N vertical, RPT x (N + 1) horizontal -- live in registers for the whole leg.
// SPAN (with PROLONG and POST == kPostRestrict): the up leg of cycle k and the down leg of cycle k + 1 in ONE pass over the
// strip -- u += P e, a.nsweep post sweeps, [the iterate of cycle k -> `out` (SPAN 1; SPAN 2 skips the store: the caller
// knows the solve cannot end at k)], sum r_k^2 of the tile -> partials, a.nsweep2 pre sweeps of cycle k + 1 -> `out2`,
// residual, full weighting.  The iterate between the two cycles is never read back: 4.75 (3.75) words per cell instead of
// 6.5.  HALO counts both sweep sets.
template <typename T, int HALO, bool PROLONG, int POST, bool ZERO_INIT, typename TX, typename TC, int SM, int W, int RPT, bool INT, int NT,
          bool VAR, int SPAN = 0>
__device__ __forceinline__ void rb_leg_body(const T* __restrict__ u, const T* __restrict__ rhs, T* __restrict__ out,
                                            const TX* __restrict__ e_coarse, TX* __restrict__ rhs_coarse,
                                            double* __restrict__ partials, const FusedArgs& a, T ihx2, T ihy2, T invD, T D, T omega,
                                            T one_m_omega, T coeff, Pack<T>* __restrict__ xbuf, TX* __restrict__ patch,
                                            double* __restrict__ red, int i0, int j0, const T* __restrict__ acoef, T sigma,
                                            const T* __restrict__ rdiag, T* __restrict__ out2 = nullptr) {
  static_assert(SPAN == 0 || (PROLONG && POST == kPostRestrict && !ZERO_INIT), "a spanning leg is an up leg followed by a down leg");
  using S = RbShape<T, HALO, W, RPT>;
  constexpr int N = S::N;
  constexpr int PH = S::RI / 2 + 2, PW = S::RJ / 2 + 2;
  constexpr int kRowPar = (1 - HALO) & 1;                           // parity of the region's first row (tiles start on odd rows, TI even)
  const int ri0 = i0 - HALO, rj0 = j0 - S::HL * N;                  // global coords of region cell (0, 0); rj0 is even
  // the wave index as a SCALAR (it is uniform, but derived from threadIdx the compiler keeps it in a vector register and turns
  // every "first / last strip row of the workgroup" test into exec-mask arithmetic): conditions on it become scalar branches
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gj0 = rj0 + lane * N;
  const int r_base = w * RPT;
  const bool col_in = INT || (gj0 >= 0 && gj0 < a.nyv);

  // ---- load: the whole strip at once -----------------------------------------------------------------------------
  const int pic0 = (ri0 >> 1) + a.ci_off, pjc0 = (rj0 >> 1) + a.cj_off;         // coarse cell of patch entry (0, 0)
  if (PROLONG) {
    for (int idx = threadIdx.x; idx < PH * PW; idx += W * 64) {
      const int pr = idx / PW, pc = idx - pr * PW;
      const int ic = pic0 + pr, jc = pjc0 + pc;
      patch[idx] = (INT || (ic >= 0 && ic < a.nxc && jc >= 0 && jc < a.nyc)) ? e_coarse[(size_t)ic * a.ldc + jc] : TX(0);
    }
  }
  Pack<T> F[RPT], U[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int gi = ri0 + r_base + k;
    F[k] = zero_pack<T>();
    U[k] = zero_pack<T>();
    if (INT || (gi >= 0 && gi < a.nx && col_in)) {
      F[k] = (NT & 4) ? ldg_nt(rhs + (size_t)gi * a.ld + gj0) : ldg(rhs + (size_t)gi * a.ld + gj0);
      if (!ZERO_INIT) U[k] = (NT & 2) ? ldg_nt(u + (size_t)gi * a.ld + gj0) : ldg(u + (size_t)gi * a.ld + gj0);
    }
  }
  // ---- VAR: face means of the lane's cells ------------------------------------------------------------------------
  constexpr int AVK = VAR ? RPT + 1 : 1, AHK = VAR ? RPT : 1, AHN = VAR ? N + 1 : 1;
  Pack<T> av[AVK];            // av[k]: faces between strip rows k - 1 and k (a(i-1/2) of row k, a(i+1/2) of row k - 1)
  T ah[AHK][AHN];             // ah[k][e]: face between the lane's cells e - 1 and e of row k (e = 0 / N: towards the neighbour lane)
  Pack<T> RD[AHK];            // RD[k]: reciprocal diagonal of the lane's cells of row k (the level's rdiag field: no division in the sweeps)
  if (VAR) {
    Pack<T> A[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int gi = ri0 + r_base + k;
      A[k] = zero_pack<T>();
      RD[VAR ? k : 0] = zero_pack<T>();
      if (INT || (gi >= 0 && gi < a.nx && col_in)) {
        A[k] = ldg(acoef + (size_t)gi * a.ld + gj0);
        RD[VAR ? k : 0] = ldg(rdiag + (size_t)gi * a.ld + gj0);
      }
    }
    Pack<T> a_above, a_below;
    rb_exchange<T, W>(xbuf + (size_t)W * 2 * 64, w, lane, A[0], A[RPT - 1], a_above, a_below);      // buffer 1: the first sweep uses buffer 0
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const Pack<T> up = (k > 0) ? A[k > 0 ? k - 1 : 0] : a_above;
      const Pack<T> dn = (k < RPT - 1) ? A[k < RPT - 1 ? k + 1 : 0] : a_below;
      const T left = dpp_from_lower_lane<T>(A[k].v[N - 1]);
      const T right = dpp_from_upper_lane<T>(A[k].v[0]);
#pragma unroll
      for (int e = 0; e < N; ++e) {
        av[VAR ? k : 0].v[e] = T(0.5) * (A[k].v[e] + up.v[e]);
        if (k == RPT - 1) av[VAR ? RPT : 0].v[e] = T(0.5) * (A[k].v[e] + dn.v[e]);
        ah[VAR ? k : 0][VAR ? e : 0] = T(0.5) * (A[k].v[e] + ((e == 0) ? left : A[k].v[e > 0 ? e - 1 : 0]));
      }
      ah[VAR ? k : 0][VAR ? N : 0] = T(0.5) * (A[k].v[N - 1] + right);
    }
  }
  if (PROLONG) {
    __syncthreads();
    using TS = typename std::conditional<(sizeof(TC) > sizeof(T)), TC, T>::type;
    if (INT) {
      // row / column parities are compile-time constants; entry (pr, pc) of the patch holds coarse (pic0 + pr, pjc0 + pc)
      constexpr int M = N / 2 + 1;
      const TX* pl = patch + (r_base / 2) * PW + lane * (N / 2);
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const bool iodd = ((kRowPar + k) & 1) != 0;                  // folds per unrolled k
        const int pr = (k + kRowPar) >> 1;                           // patch row of the coarse row at / above this fine row
        TC av[M], bv[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          av[m] = (TC)pl[pr * PW + m];
          bv[m] = iodd ? (TC)pl[(pr + 1) * PW + m] : TC(0);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const int m = e >> 1;
          TC val;
          if ((e & 1) == 0) val = iodd ? TC(0.5) * (av[m] + bv[m]) : av[m];
          else val = iodd ? TC(0.25) * (((av[m] + av[m + 1]) + bv[m]) + bv[m + 1]) : TC(0.5) * (av[m] + av[m + 1]);
          U[k].v[e] = (T)((TS)U[k].v[e] + (TS)val);
        }
      }
    } else {
      const TX* pe = patch - ((ptrdiff_t)pic0 * PW + pjc0);                       // the patch addressed like the coarse array
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int gi = ri0 + r_base + k;
        if (gi >= 0 && gi < a.nx && col_in) {
          TC val[N];
          bool ok[N];
          prolong_vec<TX, TC, N>(pe, PW, a.nxc, a.nyc, gi, gj0, a.nx, a.ny, a.sides, val, ok, a.ci_off, a.cj_off);
#pragma unroll
          for (int e = 0; e < N; ++e)
            if (ok[e]) U[k].v[e] = (T)((TS)U[k].v[e] + (TS)val[e]);
        }
      }
    }
  }

  // ---- sweeps ------------------------------------------------------------------------------------------------------
  int stage = 0;
  auto run_sweeps = [&](int nsw) {
  const int npass = (SM == kSmRbgs) ? 2 * nsw : nsw;
  for (int s = 0; s < npass; ++s, ++stage) {
    Pack<T> above, below;
    rb_exchange<T, W>(xbuf + (size_t)(stage & 1) * W * 2 * 64, w, lane, U[0], U[RPT - 1], above, below);
    const int colour = s & 1;
    // parity of (gi + gj + colour_offset) for strip row 0, cell 0: rows and cells alternate from there.  r_base = w RPT and
    // lane N are even, so the parity is the workgroup's: a scalar, and the branch on it below is not divergent
    static_assert(N % 2 == 0 && RPT % 2 == 0, "strip rows / lane cells must be even for a workgroup-uniform colour parity");
    const int par0 = (ri0 + rj0 + a.colour_offset) & 1;
    // one pass over the strip.  QP (red-black GS): compile-time parity of the pass -- cell e of strip row k is of the colour
    // being updated iff (QP + k + e) is even, so a variable-coefficient pass spends its arithmetic on those cells only
    // (constant coefficients: the packed row arithmetic computes both colours for the price of one)
    auto pass = [&](auto qp) {
      constexpr int QP = decltype(qp)::value;
      Pack<T> prev = above;                       // the (old) row above the one being updated
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = r_base + k, gi = ri0 + r;
        const Pack<T> mid = U[k];
        const Pack<T> dn = (k < RPT - 1) ? U[k + 1] : below;
        const T left = dpp_from_lower_lane<T>(mid.v[N - 1]);       // every lane active here
        const T right = dpp_from_upper_lane<T>(mid.v[0]);
        Pack<T> o = mid;
        const bool row_ok = INT ? ((k > 0 || w > 0) && (k < RPT - 1 || w < W - 1))
                                : (r >= 1 && r < S::RI - 1 && gi >= 1 && gi < a.nx - 1);
        if (row_ok && !VAR) {
          const Pack<T> res = RowMath<T, N>::relax(mid, dn, prev, left, right, F[k], ihx2, ihy2, invD, D, omega, one_m_omega, a.use_div != 0);
#pragma unroll
          for (int e = 0; e < N; ++e) {
            const int gj = gj0 + e;
            const bool mine = (SM != kSmRbgs) || (((QP + k + e) & 1) == 0);
            if ((INT || (gj >= 1 && gj < a.ny - 1)) && mine) o.v[e] = res.v[e];
          }
        }
        if (row_ok && VAR) {
#pragma unroll
          for (int e = 0; e < N; ++e) {
            if (SM == kSmRbgs && ((QP + k + e) & 1) != 0) continue;
            const T wv = (e == 0) ? left : mid.v[e > 0 ? e - 1 : 0];
            const T ea = (e == N - 1) ? right : mid.v[e < N - 1 ? e + 1 : 0];
            const T aip = av[VAR ? k + 1 : 0].v[e], aim = av[VAR ? k : 0].v[e];
            const T ajp = ah[VAR ? k : 0][VAR ? e + 1 : 0], ajm = ah[VAR ? k : 0][VAR ? e : 0];
            const T sx = aip * dn.v[e] + aim * prev.v[e], sy = ajp * ea + ajm * wv;
            const T un = (F[k].v[e] + (ihx2 * sx + ihy2 * sy)) * RD[VAR ? k : 0].v[e];
            const T res = one_m_omega * mid.v[e] + omega * un;
            const int gj = gj0 + e;
            if (INT || (gj >= 1 && gj < a.ny - 1)) o.v[e] = res;
          }
        }
        U[k] = o;
        prev = (SM == kSmRbgs) ? o : mid;         // red-black GS is in place: the neighbours it reads are not of this colour
      }
    };
    if (SM == kSmRbgs && ((par0 + colour) & 1) != 0) pass(std::integral_constant<int, 1>{});
    else pass(std::integral_constant<int, 0>{});
  }
  };
  run_sweeps(a.nsweep);

  // ---- write the tile of u' ------------------------------------------------------------------------------------------
  const bool lane_in_tile = lane >= S::HL && lane < 64 - S::HL;
  auto store_tile = [&](T* __restrict__ dst) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int r = r_base + k, gi = ri0 + r;
      if (r >= HALO && r < HALO + S::TI && lane_in_tile && (INT || (gi < a.nx && gj0 < a.nyv))) {
        if (NT & 1) stg_nt(dst + (size_t)gi * a.ld + gj0, U[k]); else stg(dst + (size_t)gi * a.ld + gj0, U[k]);
      }
    }
  };
  if (SPAN != 2) store_tile(out);
  if (POST == kPostNone) return;

  // ---- residual of the strip (r = f on boundary cells, as f is 0 outside the grid so is r) ------------------------------
  Pack<T> R[RPT];
  double acc = 0.0;
  // NORM (compile time): the residual of the tile only, its squares summed into acc; else of the whole strip (restriction)
  auto residual = [&](auto norm_stage) {
    constexpr bool NORM = decltype(norm_stage)::value;
    Pack<T> above, below;
    rb_exchange<T, W>(xbuf + (size_t)(stage & 1) * W * 2 * 64, w, lane, U[0], U[RPT - 1], above, below);
    ++stage;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int r = r_base + k, gi = ri0 + r;
      const Pack<T> up = (k > 0) ? U[k - 1] : above;
      const Pack<T> mid = U[k];
      const Pack<T> dn = (k < RPT - 1) ? U[k + 1] : below;
      const T left = dpp_from_lower_lane<T>(mid.v[N - 1]);
      const T right = dpp_from_upper_lane<T>(mid.v[0]);
      Pack<T> o = F[k];
      const bool in_tile = r >= HALO && r < HALO + S::TI && lane_in_tile;
      const bool wanted = !NORM || in_tile;                          // the norm only needs r on the tile itself
      const bool row_ok = INT ? ((k > 0 || w > 0) && (k < RPT - 1 || w < W - 1))
                              : (r >= 1 && r < S::RI - 1 && gi >= 1 && gi < a.nx - 1);
      if (wanted && row_ok && !VAR) {
        const Pack<T> rr = RowMath<T, N>::resid(mid, dn, up, left, right, F[k], ihx2, ihy2, D, coeff);
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const int gj = gj0 + e;
          if (INT || (gj >= 1 && gj < a.ny - 1)) {
            o.v[e] = rr.v[e];
            if (NORM && in_tile && (INT || (gi >= a.ni_lo && gi < a.ni_hi && gj >= a.nj_lo && gj < a.nj_hi)))
              acc += (double)o.v[e] * (double)o.v[e];
          }
        }
      }
      if (wanted && row_ok && VAR) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const T wv = (e == 0) ? left : mid.v[e > 0 ? e - 1 : 0];
          const T ea = (e == N - 1) ? right : mid.v[e < N - 1 ? e + 1 : 0];
          T au;
          if (VAR) {
            const T aip = av[VAR ? k + 1 : 0].v[e], aim = av[VAR ? k : 0].v[e];
            const T ajp = ah[VAR ? k : 0][VAR ? e + 1 : 0], ajm = ah[VAR ? k : 0][VAR ? e : 0];
            const T sx = aip * dn.v[e] + aim * up.v[e], sy = ajp * ea + ajm * wv;
            const T D0 = (aip + aim) * ihx2 + (ajp + ajm) * ihy2;
            au = coeff * ((sx * ihx2 + sy * ihy2) - mid.v[e] * ((sigma != T(0)) ? D0 + sigma : D0));
          } else {
            au = coeff * (((dn.v[e] + up.v[e]) * ihx2 + (ea + wv) * ihy2) - mid.v[e] * D);
          }
          const int gj = gj0 + e;
          if (INT || (gj >= 1 && gj < a.ny - 1)) {
            o.v[e] = F[k].v[e] - au;
            if (NORM && in_tile && (INT || (gi >= a.ni_lo && gi < a.ni_hi && gj >= a.nj_lo && gj < a.nj_hi)))
              acc += (double)o.v[e] * (double)o.v[e];
          }
        }
      }
      R[k] = o;
    }
  };
  if (POST == kPostNorm || SPAN) {
    residual(std::true_type{});
    const double t = block_reduce_sum<W>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
    if (!SPAN) return;
  }
  if (SPAN) {                       // cycle k + 1 begins: pre sweeps, the iterate the next up leg will read, then residual + restriction
    run_sweeps(a.nsweep2);
    store_tile(out2);
  }
  residual(std::false_type{});

  // ---- full weighting of the interior coarse cells that sit on this tile (operators/transfer.py:100-124) -------------
  {
    Pack<T> above, below;
    rb_exchange<T, W>(xbuf + (size_t)(stage & 1) * W * 2 * 64, w, lane, R[0], R[RPT - 1], above, below);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int r = r_base + k, fi = ri0 + r;
      const Pack<T> up = (k > 0) ? R[k - 1] : above;
      const Pack<T> mid = R[k];
      const Pack<T> dn = (k < RPT - 1) ? R[k + 1] : below;
      if (((kRowPar + k) & 1) != 0) continue;                                          // coarse rows sit on even fine rows (folds per k)
      // west neighbours of the lane's first cell, from the lane below (all lanes active)
      const T nw0 = dpp_from_lower_lane<T>(up.v[N - 1]), w0 = dpp_from_lower_lane<T>(mid.v[N - 1]), sw0 = dpp_from_lower_lane<T>(dn.v[N - 1]);
      if (r < HALO || r >= HALO + S::TI || !lane_in_tile) continue;
      const int ic = (fi >> 1) + a.ci_off;
      if (!INT && (ic < 1 || ic > a.nxc - 2 || fi < 1 || fi > a.nx - 2)) continue;
#pragma unroll
      for (int e = 0; e < N; e += 2) {                                                 // gj0 is even: e even = coarse column
        const int fj = gj0 + e, jc = (fj >> 1) + a.cj_off;
        if (!INT && (jc < 1 || jc > a.nyc - 2 || fj < 1 || fj > a.ny - 2)) continue;
        const T NWv = (e == 0) ? nw0 : up.v[e > 0 ? e - 1 : 0], Wv = (e == 0) ? w0 : mid.v[e > 0 ? e - 1 : 0],
                SWv = (e == 0) ? sw0 : dn.v[e > 0 ? e - 1 : 0];
        const T corners = ((NWv + up.v[e + 1]) + SWv) + dn.v[e + 1];
        const T edges = ((up.v[e] + dn.v[e]) + Wv) + mid.v[e + 1];
        rhs_coarse[(size_t)ic * a.ldc + jc] = (TX)((T(1.0 / 16.0) * corners + T(1.0 / 8.0) * edges) + T(1.0 / 4.0) * mid.v[e]);
      }
    }
  }
}

template <typename T, int HALO, bool PROLONG, int POST, bool ZERO_INIT, typename TX, typename TC, int TAG, int SM, int W, int RPT,
          bool VAR = false>
// VAR, fp64: the face means and the reciprocal diagonal are 44 more registers per lane; left alone the compiler takes 131-137
// in some variants and ONE 8-wave workgroup fits a CU (the up leg ran 256 us at 4097^2, 180 with two).  Four waves per
// SIMD (<= 128 registers, 4-13 of them spilled) keeps two workgroups resident.  fp32 (148-191 registers) spills 40-130
// under the same cap and loses 20-60 %: left alone.
__global__ __launch_bounds__(W * 64, (VAR && sizeof(T) == 8) ? 4 : 1) void rb_leg_kernel(
    const T* __restrict__ u, const T* __restrict__ rhs, T* __restrict__ out,
    const TX* __restrict__ e_coarse,      // PROLONG: coarse correction (dtype TX)
    TX* __restrict__ rhs_coarse,          // POST == kPostRestrict: coarse rhs (dtype TX)
    double* __restrict__ partials,        // POST == kPostNorm: one partial sum of r^2 per block
    FusedArgs a, T ihx2, T ihy2, T invD, T D, T omega, T one_m_omega, T coeff,
    const T* __restrict__ acoef,          // VAR: vertex values of the diffusion coefficient (shape / pitch of u)
    T sigma,                              // VAR: Helmholtz shift added to the per-cell diagonal
    const T* __restrict__ rdiag) {        // VAR: reciprocal diagonal per cell (var_rdiag_kernel)
  using S = RbShape<T, HALO, W, RPT>;
  constexpr int N = S::N;
  constexpr int kNT = (TAG == 2) ? MG_RB_NT_MODE : 0;           // TAG 2: the streaming-hint variant for arrays beyond the Infinity Cache
  constexpr int PH = S::RI / 2 + 2, PW = S::RJ / 2 + 2;
  constexpr size_t kXBytes = (size_t)2 * W * 2 * 64 * sizeof(Pack<T>);
  constexpr size_t kPatchBytes = PROLONG ? (size_t)PH * PW * sizeof(TX) : 0;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kXBytes + kPatchBytes];
  __shared__ double red[W];
  Pack<T>* const xbuf = reinterpret_cast<Pack<T>*>(lds);
  TX* const patch = reinterpret_cast<TX*>(lds + kXBytes);

  const int L = (a.exp_flags & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, a.ntiles);
  const int tiles_i = a.ntiles / a.tiles_j;
  const int ti = (a.exp_flags & 2) ? L % tiles_i : L / a.tiles_j, tj = (a.exp_flags & 2) ? L / tiles_i : L - ti * a.tiles_j;
  const int i0 = 1 + ti * S::TI, j0 = tj * S::TJ;
  const int ri0 = i0 - HALO, rj0 = j0 - S::HL * N;                // global coords of region cell (0, 0)
  if (a.select != 0) {
    const bool inner = ri0 >= a.in_i_lo && ri0 + S::RI <= a.in_i_hi && rj0 >= a.in_j_lo && rj0 + S::RJ <= a.in_j_hi;
    if ((a.select == 1) != inner) {
      if (POST == kPostNorm && threadIdx.x == 0) partials[blockIdx.x] = 0.0;
      return;
    }
  }
  // the region strictly inside the grid, the coarse patch inside the coarse grid, restriction targets interior coarse
  // cells, the tile inside the norm window: the guard-free body
  bool interior = !(a.exp_flags & 4) && ri0 >= 1 && ri0 + S::RI <= a.nx - 1 && rj0 >= 1 && rj0 + S::RJ <= a.ny - 1;
  if (PROLONG) {
    const int pic0 = (ri0 >> 1) + a.ci_off, pjc0 = (rj0 >> 1) + a.cj_off;
    interior = interior && pic0 >= 0 && pic0 + PH <= a.nxc && pjc0 >= 0 && pjc0 + PW <= a.nyc;
  }
  if (POST == kPostNorm) interior = interior && i0 >= a.ni_lo && i0 + S::TI <= a.ni_hi && j0 >= a.nj_lo && j0 + S::TJ <= a.nj_hi;
  if (POST == kPostRestrict)
    interior = interior && ((i0 + 1) >> 1) + a.ci_off >= 1 && ((i0 + S::TI - 1) >> 1) + a.ci_off <= a.nxc - 2 &&
               (j0 >> 1) + a.cj_off >= 1 && ((j0 + S::TJ - 2) >> 1) + a.cj_off <= a.nyc - 2;
  if (interior)
    rb_leg_body<T, HALO, PROLONG, POST, ZERO_INIT, TX, TC, SM, W, RPT, true, kNT, VAR>(u, rhs, out, e_coarse, rhs_coarse, partials, a, ihx2, ihy2,
                                                                                invD, D, omega, one_m_omega, coeff, xbuf, patch, red, i0, j0,
                                                                                acoef, sigma, rdiag);
  else
    rb_leg_body<T, HALO, PROLONG, POST, ZERO_INIT, TX, TC, SM, W, RPT, false, kNT, VAR>(u, rhs, out, e_coarse, rhs_coarse, partials, a, ihx2, ihy2,
                                                                                 invD, D, omega, one_m_omega, coeff, xbuf, patch, red, i0, j0,
                                                                                 acoef, sigma, rdiag);
}

// The spanning leg of the finest level (rb_leg_body, SPAN): up leg of cycle k + down leg of cycle k + 1.  `out_mid`: the
// iterate of cycle k (SPAN 1), `out_next`: the pre-smoothed iterate of cycle k + 1, `partials`: sum r_k^2 per workgroup.
// Whole grids only (no tile selection, no sub-array offsets beyond FusedArgs').
// Four waves per SIMD (<= 128 registers): two 8-wave workgroups per CU.  Left alone half of the variants take 129-139 and
// run one (the fp64 leg without the store in between: 171 us against 145 with it).
template <typename T, int HALO, typename TX, typename TC, int TAG, int SM, int W, int RPT, int SPAN>
__global__ __launch_bounds__(W * 64, 4) void rb_span_kernel(
    const T* __restrict__ u, const T* __restrict__ rhs, T* __restrict__ out_mid, T* __restrict__ out_next,
    const TX* __restrict__ e_coarse, TX* __restrict__ rhs_coarse, double* __restrict__ partials,
    FusedArgs a, T ihx2, T ihy2, T invD, T D, T omega, T one_m_omega, T coeff) {
  using S = RbShape<T, HALO, W, RPT>;
  constexpr int N = S::N;
  constexpr int kNT = (TAG == 2) ? MG_RB_NT_MODE : 0;
  constexpr int PH = S::RI / 2 + 2, PW = S::RJ / 2 + 2;
  constexpr size_t kXBytes = (size_t)2 * W * 2 * 64 * sizeof(Pack<T>);
  constexpr size_t kPatchBytes = (size_t)PH * PW * sizeof(TX);
  __shared__ __attribute__((aligned(16))) unsigned char lds[kXBytes + kPatchBytes];
  __shared__ double red[W];
  Pack<T>* const xbuf = reinterpret_cast<Pack<T>*>(lds);
  TX* const patch = reinterpret_cast<TX*>(lds + kXBytes);

  // Tile order: bands of a.band tile rows, numbered down the columns of a band.  Tiles that run at the same time on one
  // XCD are then vertical neighbours as well as horizontal ones, and the 2 x 6 halo rows two of them share are read from
  // HBM once and from that XCD's L2 the second time (row-major order: 586 MB per fp64 launch at 4097^2 against 470 needed)
  const int L = xcd_remap(blockIdx.x, a.ntiles);
  const int tiles_i = a.ntiles / a.tiles_j;
  const int per_band = a.band * a.tiles_j;
  const int b = L / per_band, within = L - b * per_band;
  const int rows = min(a.band, tiles_i - b * a.band);
  const int tj = within / rows, ti = b * a.band + (within - tj * rows);
  const int i0 = 1 + ti * S::TI, j0 = tj * S::TJ;
  const int ri0 = i0 - HALO, rj0 = j0 - S::HL * N;
  bool interior = ri0 >= 1 && ri0 + S::RI <= a.nx - 1 && rj0 >= 1 && rj0 + S::RJ <= a.ny - 1;
  {
    const int pic0 = (ri0 >> 1) + a.ci_off, pjc0 = (rj0 >> 1) + a.cj_off;
    interior = interior && pic0 >= 0 && pic0 + PH <= a.nxc && pjc0 >= 0 && pjc0 + PW <= a.nyc;
  }
  interior = interior && i0 >= a.ni_lo && i0 + S::TI <= a.ni_hi && j0 >= a.nj_lo && j0 + S::TJ <= a.nj_hi;
  interior = interior && ((i0 + 1) >> 1) + a.ci_off >= 1 && ((i0 + S::TI - 1) >> 1) + a.ci_off <= a.nxc - 2 &&
             (j0 >> 1) + a.cj_off >= 1 && ((j0 + S::TJ - 2) >> 1) + a.cj_off <= a.nyc - 2;
  if (interior)
    rb_leg_body<T, HALO, true, kPostRestrict, false, TX, TC, SM, W, RPT, true, kNT, false, SPAN>(
        u, rhs, out_mid, e_coarse, rhs_coarse, partials, a, ihx2, ihy2, invD, D, omega, one_m_omega, coeff, xbuf, patch, red, i0, j0,
        nullptr, T(0), nullptr, out_next);
  else
    rb_leg_body<T, HALO, true, kPostRestrict, false, TX, TC, SM, W, RPT, false, kNT, false, SPAN>(
        u, rhs, out_mid, e_coarse, rhs_coarse, partials, a, ihx2, ihy2, invD, D, omega, one_m_omega, coeff, xbuf, patch, red, i0, j0,
        nullptr, T(0), nullptr, out_next);
}

}  // namespace mg
