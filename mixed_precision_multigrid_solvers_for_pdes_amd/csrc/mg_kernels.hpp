// CDNA4 (gfx950) kernels of the geometric-multigrid V/W-cycle hot path.
//
// Layout of every field: logical shape (nx, ny), C order, index [i][j] with j
// contiguous (reference: core/grid.py:50 meshgrid 'ij'); element (i,j) lives at
// base + i*ld + j where ld (the row pitch, in elements) is a multiple of one
// 16-byte vector and base is 16-byte aligned, so every row starts on a vector
// boundary and each lane moves 16 B per access.  Columns [ny, ld) are padding:
// kernels carry them through unchanged and reductions mask them out.
//
// Stencil kernels stage a (TI+2) x (TJ+2 vectors) tile of u -- the tile plus its
// 1-cell halo ring -- through LDS, so each u element is fetched from L2/HBM once
// per tile; rhs and the output move register <-> HBM as whole vectors.  The
// blockIdx -> tile map is XCD-aware: each of the 8 XCDs sweeps a contiguous band
// of tile rows, so the vertical halo re-reads hit that XCD's own L2.
//
// Arithmetic follows the reference's association order exactly (file:line on
// each kernel); the build uses -ffp-contract=off so no FMA contraction changes
// a rounding.  Divisions by hx^2, hy^2 and the diagonal are multiplications by
// host-computed reciprocals: identical bits when h^2 is a power of two (every
// 2^k+1 grid on a unit-length domain), <= 1 ulp per operation otherwise.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace mg {

constexpr int kBlock = 256;          // 4 wave64
constexpr int kTileRowBytes = 512;   // bytes of one tile row (128 f32 / 64 f64)
constexpr int kTI = 32;              // tile rows
constexpr int kNumXcd = 8;
// TAG values of the tile kernels: same code, distinct symbols, so that a kernel trace (rocprofv3 --stats)
// reports the finest-level launches separately from the many small coarse-level ones.
constexpr int kCoarseTag = 0, kFineTag = 1;
// smoother selectors of the fused legs / coarse tail
constexpr int kSmJacobi = 0, kSmRbgs = 1;
// halo cells one full sweep consumes: 1 for Jacobi, 2 for red-black GS (one per colour pass)
__host__ __device__ constexpr int sweep_halo(int sm) { return sm == kSmRbgs ? 2 : 1; }

template <typename T> struct VecW { static constexpr int N = 16 / sizeof(T); };

template <typename T> struct alignas(16) Pack {
  T v[VecW<T>::N];
};

template <typename T> __device__ __forceinline__ Pack<T> ldg(const T* p) {
  return *reinterpret_cast<const Pack<T>*>(p);
}
template <typename T> __device__ __forceinline__ void stg(T* p, const Pack<T>& x) {
  *reinterpret_cast<Pack<T>*>(p) = x;
}
// 16-byte store with the non-temporal hint (global_store_dwordx4 ... nt): a leg's output tile is not read again before
// the whole coarser cycle has streamed through the caches
template <typename T> __device__ __forceinline__ void stg_nt(T* p, const Pack<T>& x) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(*reinterpret_cast<const v4i*>(&x), reinterpret_cast<v4i*>(p));
}
template <typename T> __device__ __forceinline__ Pack<T> ldg_nt(const T* p) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
  return *reinterpret_cast<const Pack<T>*>(&v);
}
template <typename T> __device__ __forceinline__ Pack<T> zero_pack() {
  Pack<T> z;
#pragma unroll
  for (int e = 0; e < VecW<T>::N; ++e) z.v[e] = T(0);
  return z;
}

// Bijective XCD-aware remap (blocks b and b+8 share an XCD under round-robin
// dispatch; speed only, never correctness): logical ids are dealt so that XCD k
// owns the contiguous range [k*q + min(k,r), ...).
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n / kNumXcd, r = n % kNumXcd;
  const int x = b % kNumXcd, idx = b / kNumXcd;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + idx;
}

struct TileGeom {
  int nx, ny, ld;        // logical rows, logical cols, pitch (elements)
  int nyv;               // ny rounded up to one vector: columns >= nyv are never touched
  int i_org;             // row of the first tile (0: cover the boundary rows too, 1: interior rows only)
  int tiles_j, ntiles;   // tile grid
};

template <typename T> struct TileShape {
  static constexpr int N = VecW<T>::N;
  static constexpr int TJ = kTileRowBytes / (int)sizeof(T);   // tile columns
  static constexpr int SJ = TJ + 2 * N;                       // LDS row stride (elements)
  static constexpr int VPR = SJ / N;                          // vectors per LDS row
  static constexpr int CG = TJ / N;                           // column groups (threads per tile row) = 32
  static constexpr int RG = kBlock / CG;                      // row groups = 8
  static constexpr int RPT = kTI / RG;                        // rows per thread = 4
  static constexpr int LDS_ELEMS = (kTI + 2) * SJ;
};

// Stage rows [i0-1, i0+TI] x cols [j0-N, j0+TJ+N) of `u` into LDS (zeros outside the array).
template <typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ u, T* __restrict__ s, int i0, int j0,
                                           int nx, int nyv, int ld) {
  using S = TileShape<T>;
  for (int v = threadIdx.x; v < (kTI + 2) * S::VPR; v += kBlock) {
    const int r = v / S::VPR, c = v - r * S::VPR;
    const int gi = i0 - 1 + r, gj = j0 - S::N + c * S::N;
    Pack<T> p = zero_pack<T>();
    if (gi >= 0 && gi < nx && gj >= 0 && gj < nyv) p = ldg(u + (size_t)gi * ld + gj);
    *reinterpret_cast<Pack<T>*>(s + r * S::SJ + c * S::N) = p;
  }
}

// --------------------------------------------------------------------------------------------
// Weighted Jacobi, one sweep, out of place.
//   reference: solvers/smoothers.py:62-84 (loop form), solvers/iterative.py:84-104 (vectorised):
//     nb = ihx2*(u[i+1,j]+u[i-1,j]) + ihy2*(u[i,j+1]+u[i,j-1]);  un = (rhs+nb)/D;
//     out = (1-w)*u + w*un on interior cells; boundary cells copied through.
//   DIV = false when 1/D is exact (D a power of two: square cells with dyadic h), else a true IEEE
//   division keeps the reference's rounding.
//   algorithmic traffic: read u, read rhs, write out = 3 words / DoF.
// --------------------------------------------------------------------------------------------
template <typename T, int TAG, bool DIV>
__global__ __launch_bounds__(kBlock) void jacobi_kernel(const T* __restrict__ u, const T* __restrict__ rhs,
                                                        T* __restrict__ out, TileGeom g, T ihx2, T ihy2, T invD, T D,
                                                        T omega, T one_m_omega) {
  using S = TileShape<T>;
  __shared__ __attribute__((aligned(16))) T s[S::LDS_ELEMS];
  const int L = xcd_remap(blockIdx.x, g.ntiles);
  const int ti = L / g.tiles_j, tj = L - ti * g.tiles_j;
  const int i0 = g.i_org + ti * kTI, j0 = tj * S::TJ;
  const int cg = threadIdx.x % S::CG, rg = threadIdx.x / S::CG;
  const int gj0 = j0 + cg * S::N;
  const int lr = rg * S::RPT;

  Pack<T> f[S::RPT];
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const int gi = i0 + lr + k;
    f[k] = (gi < g.nx && gj0 < g.nyv) ? ldg(rhs + (size_t)gi * g.ld + gj0) : zero_pack<T>();
  }
  stage_tile<T>(u, s, i0, j0, g.nx, g.nyv, g.ld);
  __syncthreads();

  const int lc = S::N + cg * S::N;
  Pack<T> up = *reinterpret_cast<const Pack<T>*>(s + (lr + 0) * S::SJ + lc);
  Pack<T> mid = *reinterpret_cast<const Pack<T>*>(s + (lr + 1) * S::SJ + lc);
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const Pack<T> dn = *reinterpret_cast<const Pack<T>*>(s + (lr + k + 2) * S::SJ + lc);
    const T left = s[(lr + k + 1) * S::SJ + lc - 1];
    const T right = s[(lr + k + 1) * S::SJ + lc + S::N];
    const int gi = i0 + lr + k;
    const bool row_in = (gi >= 1) && (gi < g.nx - 1);
    Pack<T> o;
#pragma unroll
    for (int e = 0; e < S::N; ++e) {
      const T w = (e == 0) ? left : mid.v[e - 1];
      const T ea = (e == S::N - 1) ? right : mid.v[e + 1];
      const T nb = ihx2 * (dn.v[e] + up.v[e]) + ihy2 * (ea + w);
      const T un = DIV ? (f[k].v[e] + nb) / D : (f[k].v[e] + nb) * invD;
      const T res = one_m_omega * mid.v[e] + omega * un;
      const int gj = gj0 + e;
      o.v[e] = (row_in && gj >= 1 && gj < g.ny - 1) ? res : mid.v[e];
    }
    if (gi < g.nx && gj0 < g.nyv) stg(out + (size_t)gi * g.ld + gj0, o);
    up = mid;
    mid = dn;
  }
}

// --------------------------------------------------------------------------------------------
// Red-black Gauss-Seidel, ONE colour per launch, in place (colour 0 = (i+j) even first).
//   reference: solvers/smoothers.py:175-207; `poff` is the parity of the global index of local
//   (0,0) so that a sub-domain keeps the global colouring.
//   Same-colour cells never neighbour each other, so in-place update within a launch is race free:
//   a launch reads only the other colour (plus its own centre value) and writes only its colour.
// --------------------------------------------------------------------------------------------
template <typename T, int TAG, bool DIV>
__global__ __launch_bounds__(kBlock) void rbgs_colour_kernel(T* __restrict__ u, const T* __restrict__ rhs, TileGeom g,
                                                             T ihx2, T ihy2, T invD, T D, T omega, T one_m_omega,
                                                             int colour, int poff) {
  using S = TileShape<T>;
  __shared__ __attribute__((aligned(16))) T s[S::LDS_ELEMS];
  const int L = xcd_remap(blockIdx.x, g.ntiles);
  const int ti = L / g.tiles_j, tj = L - ti * g.tiles_j;
  const int i0 = g.i_org + ti * kTI, j0 = tj * S::TJ;
  const int cg = threadIdx.x % S::CG, rg = threadIdx.x / S::CG;
  const int gj0 = j0 + cg * S::N;
  const int lr = rg * S::RPT;

  Pack<T> f[S::RPT];
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const int gi = i0 + lr + k;
    f[k] = (gi < g.nx && gj0 < g.nyv) ? ldg(rhs + (size_t)gi * g.ld + gj0) : zero_pack<T>();
  }
  stage_tile<T>(u, s, i0, j0, g.nx, g.nyv, g.ld);
  __syncthreads();

  const int lc = S::N + cg * S::N;
  Pack<T> up = *reinterpret_cast<const Pack<T>*>(s + (lr + 0) * S::SJ + lc);
  Pack<T> mid = *reinterpret_cast<const Pack<T>*>(s + (lr + 1) * S::SJ + lc);
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const Pack<T> dn = *reinterpret_cast<const Pack<T>*>(s + (lr + k + 2) * S::SJ + lc);
    const T left = s[(lr + k + 1) * S::SJ + lc - 1];
    const T right = s[(lr + k + 1) * S::SJ + lc + S::N];
    const int gi = i0 + lr + k;
    const bool row_in = (gi >= 1) && (gi < g.nx - 1);
    Pack<T> o;
#pragma unroll
    for (int e = 0; e < S::N; ++e) {
      const T w = (e == 0) ? left : mid.v[e - 1];
      const T ea = (e == S::N - 1) ? right : mid.v[e + 1];
      const T nb = ihx2 * (dn.v[e] + up.v[e]) + ihy2 * (ea + w);
      const T un = DIV ? (f[k].v[e] + nb) / D : (f[k].v[e] + nb) * invD;
      const T res = one_m_omega * mid.v[e] + omega * un;
      const int gj = gj0 + e;
      const bool mine = (((gi + gj + poff) & 1) == colour);
      o.v[e] = (mine && row_in && gj >= 1 && gj < g.ny - 1) ? res : mid.v[e];
    }
    if (gi < g.nx && gj0 < g.nyv) stg(u + (size_t)gi * g.ld + gj0, o);
    up = mid;
    mid = dn;
  }
}

// --------------------------------------------------------------------------------------------
// Residual r = f - A u with A = coeff * (5-point Laplacian); boundary cells r = f.
//   reference: operators/laplacian.py:73-77, 117-118:
//     Au = coeff*(( (u[i+1]+u[i-1])/hx^2 + (u[j+1]+u[j-1])/hy^2 ) - u*(2/hx^2+2/hy^2))
//   WRITE_R: store r (3 words/DoF) ; NORM: also emit one fp64 partial sum of r^2 per block
//   (wave64 shuffle reduction, then 4 waves through LDS) -> ||r||^2 needs no second pass over r.
// --------------------------------------------------------------------------------------------
// Sum over the wave, valid in lane 0.  The tree is x[l] += x[l + off] for off = 32, 16, 8, 4, 2, 1 -- what a
// __shfl_down loop computes -- but without its six dependent ds_bpermute round trips (~130 cycles each, twice per
// double): the two cross-row steps are gfx950's v_permlane32_swap / v_permlane16_swap, the four in-row steps DPP
// row_shl.  The coarsest-level stop test runs this once per Gauss-Seidel sweep, ~40 times per cycle.
template <int CTRL>
__device__ __forceinline__ double dpp_row_move(double x) {
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true),
                          __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double wave_reduce_sum(double x) {
  {   // lanes 0..31 += lanes 32..63   (swap: lanes [32:63] of the first operand <-> lanes [0:31] of the second)
    const int lo = __double2loint(x), hi = __double2hiint(x);
    x += __hiloint2double(__builtin_amdgcn_permlane32_swap(hi, hi, false, false)[1],
                          __builtin_amdgcn_permlane32_swap(lo, lo, false, false)[1]);
  }
  {   // rows 0 / 2 += rows 1 / 3        (swap: odd rows of the first operand <-> even rows of the second)
    const int lo = __double2loint(x), hi = __double2hiint(x);
    x += __hiloint2double(__builtin_amdgcn_permlane16_swap(hi, hi, false, false)[1],
                          __builtin_amdgcn_permlane16_swap(lo, lo, false, false)[1]);
  }
  x += dpp_row_move<0x108>(x);      // row_shl:8
  x += dpp_row_move<0x104>(x);      // row_shl:4
  x += dpp_row_move<0x102>(x);      // row_shl:2
  x += dpp_row_move<0x101>(x);      // row_shl:1
  return x;
}
// The same tree restricted to row 0 (lanes 0..15): exact when lanes 16..63 hold zeros -- the two cross-row steps of
// wave_reduce_sum then add 0.0 -- so the result has the same bits.
__device__ __forceinline__ double row0_reduce_sum(double x) {
  x += dpp_row_move<0x108>(x);      // row_shl:8
  x += dpp_row_move<0x104>(x);      // row_shl:4
  x += dpp_row_move<0x102>(x);      // row_shl:2
  x += dpp_row_move<0x101>(x);      // row_shl:1
  return x;
}
// lane 0's value in every lane (v_readfirstlane: all lanes are active where this is used)
__device__ __forceinline__ double wave_first(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// Sum over the block of NW waves; result valid in thread 0.  `red` is >= NW doubles of LDS.
template <int NW = kBlock / 64>
__device__ __forceinline__ double block_reduce_sum(double x, double* red) {
  x = wave_reduce_sum(x);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = x;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w];
  }
  return t;
}

template <typename T, bool WRITE_R, bool NORM, int TAG>
__global__ __launch_bounds__(kBlock) void residual_kernel(const T* __restrict__ u, const T* __restrict__ rhs,
                                                          T* __restrict__ r, double* __restrict__ partials, TileGeom g,
                                                          T ihx2, T ihy2, T diag, T coeff) {
  using S = TileShape<T>;
  __shared__ __attribute__((aligned(16))) T s[S::LDS_ELEMS];
  __shared__ double red[kBlock / 64];
  const int L = xcd_remap(blockIdx.x, g.ntiles);
  const int ti = L / g.tiles_j, tj = L - ti * g.tiles_j;
  const int i0 = g.i_org + ti * kTI, j0 = tj * S::TJ;
  const int cg = threadIdx.x % S::CG, rg = threadIdx.x / S::CG;
  const int gj0 = j0 + cg * S::N;
  const int lr = rg * S::RPT;

  Pack<T> f[S::RPT];
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const int gi = i0 + lr + k;
    f[k] = (gi < g.nx && gj0 < g.nyv) ? ldg(rhs + (size_t)gi * g.ld + gj0) : zero_pack<T>();
  }
  stage_tile<T>(u, s, i0, j0, g.nx, g.nyv, g.ld);
  __syncthreads();

  const int lc = S::N + cg * S::N;
  Pack<T> up = *reinterpret_cast<const Pack<T>*>(s + (lr + 0) * S::SJ + lc);
  Pack<T> mid = *reinterpret_cast<const Pack<T>*>(s + (lr + 1) * S::SJ + lc);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const Pack<T> dn = *reinterpret_cast<const Pack<T>*>(s + (lr + k + 2) * S::SJ + lc);
    const T left = s[(lr + k + 1) * S::SJ + lc - 1];
    const T right = s[(lr + k + 1) * S::SJ + lc + S::N];
    const int gi = i0 + lr + k;
    const bool row_in = (gi >= 1) && (gi < g.nx - 1);
    Pack<T> o;
#pragma unroll
    for (int e = 0; e < S::N; ++e) {
      const T w = (e == 0) ? left : mid.v[e - 1];
      const T ea = (e == S::N - 1) ? right : mid.v[e + 1];
      const T au = coeff * (((dn.v[e] + up.v[e]) * ihx2 + (ea + w) * ihy2) - mid.v[e] * diag);
      const int gj = gj0 + e;
      const bool interior = row_in && gj >= 1 && gj < g.ny - 1;
      const T rv = interior ? (f[k].v[e] - au) : f[k].v[e];
      o.v[e] = rv;
      if (NORM && gi < g.nx && gj < g.ny) acc += (double)rv * (double)rv;
    }
    if (WRITE_R && gi < g.nx && gj0 < g.ld) stg(r + (size_t)gi * g.ld + gj0, o);
    up = mid;
    mid = dn;
  }
  if (NORM) {
    const double t = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// --------------------------------------------------------------------------------------------
// Residual across precisions: r (TR) = f (TF) - A (u (TU) [+ e (float)]), evaluated in double.
//   (a) TU = TF = float, TR = double: the reference's MixedPrecisionKernels.compute_mixed_precision_residual
//       (gpu/cuda_kernels.py:843-883, 937-967: fp32 iterate and rhs in, fp64 residual out, 16 B / DoF) with the oracle's
//       operator and boundary convention (A = coeff * Laplacian, r = f on boundary cells) instead of that kernel's
//       inconsistent ones;
//   (b) TU = TF = double, TR = float, UPDATE: one outer step of defect correction (MG_PREC_DEFECT): u' = u + e is formed
//       on the fly (written to `u_out`), r = f - A u' becomes the fp32 right-hand side of the next error equation
//       (ZERO_RING: 0 on boundary cells -- the Dirichlet data of u are exact, the error vanishes there) and
//       sum r^2 (boundary cells: f^2, the reference's norm, operators/laplacian.py:117-118) goes to `partials`.
//   Each thread produces two adjacent cells of one row; neighbours come straight from L1 / L2 (a side path: one pass
//   per outer iteration, not a smoother).
// --------------------------------------------------------------------------------------------
template <typename TU, typename TF, typename TR, bool UPDATE, bool ZERO_RING, bool NORM>
__global__ __launch_bounds__(kBlock) void residual_xprec_kernel(const TU* __restrict__ u, const float* __restrict__ e,
                                                                const TF* __restrict__ f, TR* __restrict__ r,
                                                                double* __restrict__ u_out, double* __restrict__ partials, int nx,
                                                                int ny, int ldu, int lde, int ldf, int ldr, double ihx2, double ihy2,
                                                                double diag, double coeff) {
  __shared__ double red[kBlock / 64];
  const int pairs = (ny + 1) / 2;
  const long long total = (long long)nx * pairs;
  double acc = 0.0;
  auto val = [&](int i, int j) -> double {
    double x = (double)u[(size_t)i * ldu + j];
    if (UPDATE) x += (double)e[(size_t)i * lde + j];
    return x;
  };
  for (long long v = (long long)blockIdx.x * kBlock + threadIdx.x; v < total; v += (long long)gridDim.x * kBlock) {
    const int i = (int)(v / pairs), j0 = (int)(v - (long long)i * pairs) * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = j0 + q;
      if (j >= ny) continue;
      const double fv = (double)f[(size_t)i * ldf + j];
      const double c = val(i, j);
      double rv;
      if (i >= 1 && i < nx - 1 && j >= 1 && j < ny - 1) {
        const double au = coeff * (((val(i + 1, j) + val(i - 1, j)) * ihx2 + (val(i, j + 1) + val(i, j - 1)) * ihy2) - c * diag);
        rv = fv - au;
        r[(size_t)i * ldr + j] = (TR)rv;
      } else {
        rv = fv;
        r[(size_t)i * ldr + j] = ZERO_RING ? TR(0) : (TR)fv;
      }
      if (UPDATE) u_out[(size_t)i * ldu + j] = c;
      if (NORM) acc += rv * rv;
    }
  }
  if (NORM) {
    const double t = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// Sum of squares of field[i_lo:i_hi, j_lo:j_hi] (reference: core/grid.py:187 without the hx*hy factor and
// sqrt).  The window lets a sub-domain count exactly the cells it owns (plus its physical boundary).
template <typename T>
__global__ __launch_bounds__(kBlock) void sumsq_kernel(const T* __restrict__ x, double* __restrict__ partials, int ld,
                                                       int i_lo, int i_hi, int j_lo, int j_hi) {
  constexpr int N = VecW<T>::N;
  __shared__ double red[kBlock / 64];
  const int c_lo = j_lo / N, c_hi = (j_hi + N - 1) / N;
  const int vpr = c_hi - c_lo;
  const long long total = (long long)(i_hi - i_lo) * vpr;
  double acc = 0.0;
  for (long long v = (long long)blockIdx.x * kBlock + threadIdx.x; v < total; v += (long long)gridDim.x * kBlock) {
    const int i = i_lo + (int)(v / vpr), c = c_lo + (int)(v % vpr);
    const Pack<T> p = ldg(x + (size_t)i * ld + c * N);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const int j = c * N + e;
      if (j >= j_lo && j < j_hi) acc += (double)p.v[e] * (double)p.v[e];
    }
  }
  const double t = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// Host-visible mailbox for one scalar: the kernel stores the value, fences at system scope, then stores the
// sequence number; the host spins on `seq` (pinned, coherent host memory) instead of paying a stream
// synchronisation per multigrid iteration.
struct HostMailbox {
  double value;
  unsigned long long seq;
};

// Fixed-order final reduction of the per-block partials (deterministic, one block).  It sits between two cycles on
// the stream, so it is built for latency: 1024 threads, four independent 16-byte loads in flight per thread.
constexpr int kReduceBlock = 1024;
template <int TAG>       // a template only so that the header can be included by several translation units
__global__ __launch_bounds__(kReduceBlock) void reduce_partials_kernel(const double* __restrict__ partials, int n,
                                                                       double* __restrict__ out, HostMailbox* mailbox,
                                                                       unsigned long long seq) {
  __shared__ double red[kReduceBlock / 64];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int nv = n >> 1;                                  // whole double2 vectors (the buffer is 16-byte aligned)
  const Pack<double>* pv = reinterpret_cast<const Pack<double>*>(partials);
  int i = threadIdx.x;
  for (; i + 3 * kReduceBlock < nv; i += 4 * kReduceBlock) {
    const Pack<double> x0 = pv[i], x1 = pv[i + kReduceBlock], x2 = pv[i + 2 * kReduceBlock], x3 = pv[i + 3 * kReduceBlock];
    a0 += x0.v[0] + x0.v[1]; a1 += x1.v[0] + x1.v[1]; a2 += x2.v[0] + x2.v[1]; a3 += x3.v[0] + x3.v[1];
  }
  for (; i < nv; i += kReduceBlock) { const Pack<double> x = pv[i]; a0 += x.v[0] + x.v[1]; }
  if ((n & 1) && threadIdx.x == 0) a1 += partials[n - 1];
  const double t = block_reduce_sum<kReduceBlock / 64>((a0 + a1) + (a2 + a3), red);
  if (threadIdx.x == 0) {
    *out = t;
    if (mailbox) {
      mailbox->value = t;
      __threadfence_system();
      __hip_atomic_store(&mailbox->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// --------------------------------------------------------------------------------------------
// Full-weighting restriction, fine (nxf,nyf) -> coarse (nxc,nyc); coarse (ic,jc) sits on fine (2ic,2jc).
//   reference: operators/transfer.py:100-124 -- interior:
//     1/16*(((NW+NE)+SW)+SE) + 1/8*(((N+S)+W)+E) + 1/4*C   evaluated in the FIELD's dtype,
//     result stored in the coarse grid's dtype; coarse boundary = injection.
//   `sides` says which edges of this array are PHYSICAL boundaries (bit 0: i = 0, 1: i = nxc-1,
//   2: j = 0, 3: j = nyc-1).  A whole grid has all four; on a sub-domain the other edges are ghost
//   rings owned by a neighbour: those coarse cells are left untouched (filled by the halo exchange).
//   One thread produces one vector of coarse cells of one coarse row.
// --------------------------------------------------------------------------------------------
constexpr int kSideILo = 1, kSideIHi = 2, kSideJLo = 4, kSideJHi = 8, kAllSides = 15;

template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void restrict_fw_kernel(const TIN* __restrict__ fine, TOUT* __restrict__ coarse,
                                                             int ldf, int nxc, int nyc, int ldc, int sides) {
  constexpr int NO = VecW<TOUT>::N;
  const int vpr = (nyc + NO - 1) / NO;
  const long long total = (long long)nxc * vpr;
  for (long long v = (long long)blockIdx.x * kBlock + threadIdx.x; v < total; v += (long long)gridDim.x * kBlock) {
    const int ic = (int)(v / vpr), c = (int)(v - (long long)ic * vpr);
    const int jc0 = c * NO;
    const int fi = 2 * ic;
    Pack<TOUT> o = ldg(coarse + (size_t)ic * ldc + jc0);
#pragma unroll
    for (int e = 0; e < NO; ++e) {
      const int jc = jc0 + e, fj = 2 * jc;
      if (jc >= nyc) continue;
      const bool edge_i = (ic == 0) || (ic == nxc - 1), edge_j = (jc == 0) || (jc == nyc - 1);
      const TIN* p = fine + (size_t)fi * ldf + fj;
      if (!edge_i && !edge_j) {
        const TIN corners = ((p[-ldf - 1] + p[-ldf + 1]) + p[ldf - 1]) + p[ldf + 1];
        const TIN edges = ((p[-ldf] + p[ldf]) + p[-1]) + p[1];
        o.v[e] = (TOUT)((TIN(1.0 / 16.0) * corners + TIN(1.0 / 8.0) * edges) + TIN(1.0 / 4.0) * p[0]);
      } else {
        const bool ghost = (ic == 0 && !(sides & kSideILo)) || (ic == nxc - 1 && !(sides & kSideIHi)) ||
                           (jc == 0 && !(sides & kSideJLo)) || (jc == nyc - 1 && !(sides & kSideJHi));
        if (!ghost) o.v[e] = (TOUT)p[0];     // physical boundary: injection
      }
    }
    stg(coarse + (size_t)ic * ldc + jc0, o);
  }
}

// Interpolated correction for the N fine cells (gi, gj0 .. gj0+N-1), gj0 even: the <= N/2+1 coarse values of
// each of the two coarse rows are loaded once and shared by the cells (column parity is known at compile time).
//   reference: operators/transfer.py:239-265; far-edge zeros (F9) on physical far edges (`sides`).
template <typename TX, typename TC, int N>
__device__ __forceinline__ void prolong_vec(const TX* __restrict__ e, int ldc, int nxc, int nyc, int gi, int gj0,
                                            int nxf, int nyf, int sides, TC (&val)[N], bool (&ok)[N], int ci_off = 0,
                                            int cj_off = 0) {
  // ci_off / cj_off: coarse cell (ic, jc) sits on fine cell (2 (ic - ci_off), 2 (jc - cj_off)) -- non-zero on
  // sub-domains whose coarse array carries a wider ghost zone (in fine units) than the fine one
  constexpr int M = N / 2 + 1;
  const int ic = (gi >> 1) + ci_off, jc0 = (gj0 >> 1) + cj_off;
  const bool iodd = gi & 1;
  const bool row_ok = ic >= 0 && (ic + (iodd ? 1 : 0)) < nxc;
  const TX* r0 = e + (size_t)(row_ok ? ic : 0) * ldc + jc0;
  const TX* r1 = r0 + ((iodd && row_ok) ? ldc : 0);
  TC a[M], b[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const bool in = row_ok && (jc0 + m >= 0) && (jc0 + m < nyc);
    a[m] = in ? (TC)r0[m] : TC(0);
    b[m] = in ? (TC)r1[m] : TC(0);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int j = gj0 + k, m = k >> 1;
    if ((k & 1) == 0) {
      ok[k] = row_ok && (j < nyf) && (jc0 + m >= 0) && (jc0 + m < nyc);
      val[k] = iodd ? (((sides & kSideJHi) && j == nyf - 1) ? TC(0) : TC(0.5) * (a[m] + b[m])) : a[m];
    } else {
      ok[k] = row_ok && (j < nyf) && (jc0 + m >= 0) && (jc0 + m + 1 < nyc);
      val[k] = iodd ? TC(0.25) * (((a[m] + a[m + 1]) + b[m]) + b[m + 1])
                    : (((sides & kSideIHi) && gi == nxf - 1) ? TC(0) : TC(0.5) * (a[m] + a[m + 1]));
    }
  }
}

// --------------------------------------------------------------------------------------------
// Bilinear prolongation fused with the correction:  u += P e   (or u = P e when ADD == false).
//   reference: operators/transfer.py:234-267 + solvers/multigrid.py:329.  Interpolation is evaluated
//   in TC (the fine GRID's dtype in the reference), the sum u + Pe in the wider of (TF, TC), then
//   rounded to TF -- NumPy's `u += fine_correction` semantics.
//   Quirk F9 reproduced on PHYSICAL far edges (sides bits 1 and 3): (odd i, j == ny-1) and
//   (i == nx-1, odd j) receive 0.  On a sub-domain the far edge may be a ghost ring instead, which is
//   interpolated like any other cell from the (exchanged) coarse ghost values; fine cells whose coarse
//   partners fall outside the coarse array (nxc, nyc) are left untouched.
// --------------------------------------------------------------------------------------------
template <typename TC_IN, typename TF, typename TC, bool ADD>
__global__ __launch_bounds__(kBlock) void prolong_kernel(const TC_IN* __restrict__ e, TF* __restrict__ u, int nxf,
                                                         int nyf, int ldf, int nxc, int nyc, int ldc, int sides) {
  constexpr int N = VecW<TF>::N;
  using TS = typename std::conditional<(sizeof(TC) > sizeof(TF)), TC, TF>::type;
  const int vpr = (nyf + N - 1) / N;
  const long long total = (long long)nxf * vpr;
  for (long long v = (long long)blockIdx.x * kBlock + threadIdx.x; v < total; v += (long long)gridDim.x * kBlock) {
    const int i = (int)(v / vpr), c = (int)(v - (long long)i * vpr);
    const int j0 = c * N;
    TC val[N];
    bool ok[N];
    prolong_vec<TC_IN, TC, N>(e, ldc, nxc, nyc, i, j0, nxf, nyf, sides, val, ok);
    Pack<TF> uo = ldg(u + (size_t)i * ldf + j0);
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (ok[k]) uo.v[k] = ADD ? (TF)((TS)uo.v[k] + (TS)val[k]) : (TF)val[k];
    stg(u + (size_t)i * ldf + j0, uo);
  }
}

// u[1:nx-1, 1:ny-1] = 0, boundary ring kept (the full-multigrid start on the finest level: Dirichlet data stay).
template <typename T>
__global__ __launch_bounds__(kBlock) void zero_interior_kernel(T* __restrict__ u, int nx, int ny, int ld) {
  constexpr int N = VecW<T>::N;
  const int vpr = (ny + N - 1) / N;
  const long long total = (long long)(nx - 2) * vpr;
  for (long long v = (long long)blockIdx.x * kBlock + threadIdx.x; v < total; v += (long long)gridDim.x * kBlock) {
    const int i = 1 + (int)(v / vpr), j0 = (int)(v % vpr) * N;
    Pack<T> p = ldg(u + (size_t)i * ld + j0);
#pragma unroll
    for (int e = 0; e < N; ++e)
      if (j0 + e >= 1 && j0 + e < ny - 1) p.v[e] = T(0);
    stg(u + (size_t)i * ld + j0, p);
  }
}

// Bare 2-read + 1-write stream in the tile shape of the register-blocked legs (4 waves x 4 rows x 1 KB per workgroup, all
// loads issued before the first store): c = a + b.  Not part of the path -- the yardstick bench.py times next to the
// Jacobi sweep (mg_time_op op 11): what the memory system delivers for the sweep's traffic shape with no stencil at all.
template <typename T>
__global__ __launch_bounds__(256) void stream_triad_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ c, int nx,
                                                           int nyv, int ld, int tiles_j) {
  constexpr int N = VecW<T>::N, RPT = 4, W = 4;          // 16 rows x 1 KB: the fastest bare shape measured (tools/stream_pattern_bench2.hip)
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = (tj * 64 + lane) * N, i0 = ti * (W * RPT) + w * RPT;
  Pack<T> x[RPT], y[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    x[k] = zero_pack<T>(); y[k] = zero_pack<T>();
    if (i0 + k < nx && j < nyv) { x[k] = ldg(a + (size_t)(i0 + k) * ld + j); y[k] = ldg(b + (size_t)(i0 + k) * ld + j); }
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    if (i0 + k < nx && j < nyv) {
      Pack<T> o;
#pragma unroll
      for (int e = 0; e < N; ++e) o.v[e] = x[k].v[e] + y[k].v[e];
      stg(c + (size_t)(i0 + k) * ld + j, o);
    }
  }
}

// Element-wise precision switch (reference: core/precision.py:106-134 `astype`).
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void convert_kernel(const TIN* __restrict__ in, TOUT* __restrict__ out, int nx,
                                                         int ny, int ldi, int ldo) {
  const long long total = (long long)nx * ny;
  for (long long v = (long long)blockIdx.x * kBlock + threadIdx.x; v < total; v += (long long)gridDim.x * kBlock) {
    const int i = (int)(v / ny), j = (int)(v - (long long)i * ny);
    out[(size_t)i * ldo + j] = (TOUT)in[(size_t)i * ldi + j];
  }
}

// The boundary ring only (rows 0 and nx-1, columns 0 and ny-1): what the ping-pong partner of an iterate needs -- the
// fused legs rewrite its interior and keep its ring.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void convert_ring_kernel(const TIN* __restrict__ in, TOUT* __restrict__ out, int nx,
                                                              int ny, int ldi, int ldo) {
  const int total = 2 * ny + 2 * nx;
  for (int v = blockIdx.x * kBlock + threadIdx.x; v < total; v += gridDim.x * kBlock) {
    int i, j;
    if (v < ny) { i = 0; j = v; }
    else if (v < 2 * ny) { i = nx - 1; j = v - ny; }
    else if (v < 2 * ny + nx) { i = v - 2 * ny; j = 0; }
    else { i = v - 2 * ny - nx; j = ny - 1; }
    out[(size_t)i * ldo + j] = (TOUT)in[(size_t)i * ldi + j];
  }
}

// --------------------------------------------------------------------------------------------
// Coarsest-grid solver: lexicographic Gauss-Seidel sweeps until sqrt(hx*hy*sum r^2) < tol or maxit.
//   reference: solvers/smoothers.py:153-173 driven by IterativeSolver.solve solvers/base.py:255-290
//   (coarse_tolerance 1e-12, coarse_max_iterations 1000: solvers/multigrid.py:119-124).
//   One workgroup.  Cells of one anti-diagonal i+j = s depend only on diagonal s-1 (already new) and
//   s+1 (still old), so sweeping diagonals in order with a barrier in between reproduces the
//   lexicographic loop bit for bit.  Every wave leaves through the same uniform exit test.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void coarse_lexgs_kernel(T* __restrict__ u, const T* __restrict__ rhs, int nx,
                                                              int ny, int ld, T hx2, T hy2, T omega,
                                                              T one_m_omega, T diag, T coeff, double hxhy, double tol,
                                                              int maxit, int* __restrict__ sweeps_out,
                                                              const T* __restrict__ a = nullptr, T sigma = T(0)) {
  __shared__ double red[kBlock / 64];
  __shared__ double total;
  int it = 0;
  for (it = 1; it <= maxit; ++it) {
    for (int sdiag = 2; sdiag <= nx + ny - 4; ++sdiag) {
      const int ilo = max(1, sdiag - (ny - 2)), ihi = min(nx - 2, sdiag - 1);
      for (int i = ilo + (int)threadIdx.x; i <= ihi; i += kBlock) {
        const int j = sdiag - i;
        T* p = u + (size_t)i * ld + j;
        if (a) {        // variable coefficient: face means of a, per-cell diagonal
          const T* q = a + (size_t)i * ld + j;
          const T aip = T(0.5) * (q[0] + q[ld]), aim = T(0.5) * (q[0] + q[-ld]), ajp = T(0.5) * (q[0] + q[1]), ajm = T(0.5) * (q[0] + q[-1]);
          const T nb = (aip * p[ld] + aim * p[-ld]) / hx2 + (ajp * p[1] + ajm * p[-1]) / hy2;
          const T D = (aip + aim) / hx2 + (ajp + ajm) / hy2 + sigma;
          const T un = (rhs[(size_t)i * ld + j] + nb) * (T(1) / D);          // variable coefficient: times the reciprocal diagonal
          p[0] = one_m_omega * p[0] + omega * un;
          continue;
        }
        const T nb = (p[ld] + p[-ld]) / hx2 + (p[1] + p[-1]) / hy2;
        const T un = (rhs[(size_t)i * ld + j] + nb) / diag;
        p[0] = one_m_omega * p[0] + omega * un;
      }
      __syncthreads();
    }
    double acc = 0.0;
    for (int idx = threadIdx.x; idx < nx * ny; idx += kBlock) {
      const int i = idx / ny, j = idx - i * ny;
      const T* p = u + (size_t)i * ld + j;
      T rv = rhs[(size_t)i * ld + j];
      if (i >= 1 && i < nx - 1 && j >= 1 && j < ny - 1) {
        if (a) {
          const T* q = a + (size_t)i * ld + j;
          const T aip = T(0.5) * (q[0] + q[ld]), aim = T(0.5) * (q[0] + q[-ld]), ajp = T(0.5) * (q[0] + q[1]), ajm = T(0.5) * (q[0] + q[-1]);
          const T D = (aip + aim) / hx2 + (ajp + ajm) / hy2 + sigma;
          rv = rv - coeff * (((aip * p[ld] + aim * p[-ld]) / hx2 + (ajp * p[1] + ajm * p[-1]) / hy2) - p[0] * D);
        } else {
          rv = rv - coeff * (((p[ld] + p[-ld]) / hx2 + (p[1] + p[-1]) / hy2) - p[0] * diag);
        }
      }
      acc += (double)rv * (double)rv;
    }
    const double t = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) total = t;
    __syncthreads();
    const bool done = sqrt(hxhy * total) < tol;
    __syncthreads();
    if (done) break;
  }
  if (threadIdx.x == 0 && sweeps_out) *sweeps_out = (it > maxit) ? maxit : it;
}


// --------------------------------------------------------------------------------------------
// Lexicographic Gauss-Seidel, software-pipelined ACROSS sweeps, for tiny grids (one wave).
//   Cell (i,j) of sweep k needs the sweep-k values of (i-1,j),(i,j-1) and the sweep-(k-1) values of
//   (i+1,j),(i,j+1).  Give cell c on anti-diagonal d = i+j the time slot t = 2k + d: at time t every cell with
//   (t - d) even is updated IN PLACE from its four neighbours' current values -- they are exactly the
//   required sweep-k / sweep-(k-1) values, because neighbours sit on diagonals d-1 / d+1 and were updated
//   at t-1.  One sweep then costs two time steps instead of nx+ny-5, with bit-identical arithmetic.
//   The reference tests ||r|| after every complete sweep (solvers/base.py:271-283), so each cell also logs
//   its sweep-k value into a ring of H snapshots; sweep k is complete at t = 2k + dmax, its residual norm is
//   evaluated from snapshot k, and when it meets the tolerance (or k == maxit) snapshot k IS the result
//   (the few speculative updates of later sweeps are dropped).  Returns the number of sweeps.
// --------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T rhx2_of(T h2) { return T(1) / h2; }
constexpr int kPipeCells = 81;     // up to 9 x 9
constexpr int kPipeSlots = 9;      // >= (dmax - 2) / 2 + 2 for 9 x 9 (dmax = 14)

template <typename T>
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int CTRL> __device__ __forceinline__ float dpp_row_move(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}

// The 5 x 5 coarsest grid (3 x 3 unknowns -- the end of every hierarchy of 2^k + 1 grids): the same pipelined sweeps
// with the nine unknowns in lanes 0..8 and NO LDS traffic in the loop.  Neighbours come from lanes c -+ 1, c -+ 3 by
// DPP row shifts (ring neighbours are per-lane constants); a lane keeps its last three sweep values in a shift
// register, and because a lane on diagonal d is always floor((6 - d) / 2) sweeps ahead of the sweep under test, the
// snapshot the stop test needs sits at a per-lane constant depth.  Same expressions, same sweep / stop semantics as
// the LDS version below (the stop test sums its nine squares in a different lane order).
// VAR: variable coefficient (vertex values in `sa`, LDS): every lane keeps the four face means and the diagonal of its
// cell in registers; the arithmetic is coarse_lexgs_kernel's variable-coefficient branch.
template <typename T, bool VAR = false>
__device__ int lexgs_pipelined_5x5(T* __restrict__ su, const T* __restrict__ sf, T hx2, T hy2, T diag, T coeff, T omega,
                                   T one_m_omega, bool exact, double hxhy, double tol_x, int maxit, int lane,
                                   const T* __restrict__ sa = nullptr, T sigma = T(0)) {
  // tol_x: the stop test sqrt(hx hy sum r^2) < tol as  hx hy sum r^2 < tol_x  with tol_x = min{x : sqrt(x) >= tol}
  // (host, sqrt_threshold): the same decision for every x -- IEEE sqrt is monotone -- without the ~20 dependent
  // instructions of a double-precision square root in every sweep of a one-wave latency chain
  constexpr int ny = 5;
  const T rhx2 = T(1) / hx2, rhy2 = T(1) / hy2, rdiag = T(1) / diag;
  double ring = 0.0;
  if (lane < 25) {
    const int i = lane / ny, j = lane - i * ny;
    if (i == 0 || i == 4 || j == 0 || j == 4) ring = (double)sf[lane] * (double)sf[lane];
  }
  ring = wave_first(wave_reduce_sum(ring));
  const bool mine = lane < 9;
  const int ci = mine ? lane / 3 : 0, cj = mine ? lane - 3 * (lane / 3) : 0;
  const int gi = ci + 1, gj = cj + 1, g = gi * ny + gj, d = gi + gj;
  const T fv = mine ? sf[g] : T(0);
  T uv = mine ? su[g] : T(0);
  // ring neighbours (constant over the solve); interior neighbours are fetched from the adjacent lanes every step
  const bool up_ring = gi == 1, dn_ring = gi == 3, lf_ring = gj == 1, rt_ring = gj == 3;
  const T up_c = su[g - ny], dn_c = su[g + ny], lf_c = su[g - 1], rt_c = su[g + 1];
  T aip = T(1), aim = T(1), ajp = T(1), ajm = T(1), Dv = diag;
  if (VAR) {
    const T ac = sa[g];
    aip = T(0.5) * (ac + sa[g + ny]); aim = T(0.5) * (ac + sa[g - ny]); ajp = T(0.5) * (ac + sa[g + 1]); ajm = T(0.5) * (ac + sa[g - 1]);
    Dv = (exact ? (aip + aim) * rhx2 + (ajp + ajm) * rhy2 : (aip + aim) / hx2 + (ajp + ajm) / hy2) + sigma;
  }
  const T rDv = T(1) / Dv;                        // VAR: the smoothers multiply by the reciprocal diagonal (oracle: _var_update)
  T h0 = uv, h1 = uv, h2 = uv;                    // the lane's last three sweep values, newest first
  int tnext = 2 + d, knext = 1;
  int kc = 1, tdone = 2 + 6;
  const int lead = (6 - d) >> 1;                  // sweeps this lane is ahead of the sweep under test
  int sweeps = maxit;
  T result = uv;
  for (int t = 4;; ++t) {
    {
      // the four moves run with EVERY lane active (a DPP read of a lane that is masked off returns 0), the choice
      // between ring constant and neighbour comes afterwards
      const T n_up = dpp_row_move<0x113>(uv);     // row_shr:3  (lane c - 3)
      const T n_dn = dpp_row_move<0x103>(uv);     // row_shl:3  (lane c + 3)
      const T n_lf = dpp_row_move<0x111>(uv);     // row_shr:1
      const T n_rt = dpp_row_move<0x101>(uv);     // row_shl:1
      const T up = up_ring ? up_c : n_up, dn = dn_ring ? dn_c : n_dn, lf = lf_ring ? lf_c : n_lf, rt = rt_ring ? rt_c : n_rt;
      const T sx = VAR ? aip * dn + aim * up : dn + up, sy = VAR ? ajp * rt + ajm * lf : rt + lf;
      const T nb = exact ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2;
      const T num = fv + nb;
      const T un = VAR ? num * rDv : (exact ? num * rdiag : num / diag);
      const T nv = one_m_omega * uv + omega * un;
      if (mine && t == tnext && knext <= maxit) {
        uv = nv;
        h2 = h1; h1 = h0; h0 = nv;
        tnext += 2;
        ++knext;
      }
    }
    if (t == tdone) {
      const int depth = min(lead, maxit - kc);
      const T snap = depth == 0 ? h0 : (depth == 1 ? h1 : h2);
      const T n_up = dpp_row_move<0x113>(snap), n_dn = dpp_row_move<0x103>(snap);
      const T n_lf = dpp_row_move<0x111>(snap), n_rt = dpp_row_move<0x101>(snap);
      const T up = up_ring ? up_c : n_up, dn = dn_ring ? dn_c : n_dn, lf = lf_ring ? lf_c : n_lf, rt = rt_ring ? rt_c : n_rt;
      const T sx = VAR ? aip * dn + aim * up : dn + up, sy = VAR ? ajp * rt + ajm * lf : rt + lf;
      const T rv = fv - coeff * ((exact ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2) - snap * (VAR ? Dv : diag));
      double acc = mine ? (double)rv * (double)rv : 0.0;
      acc = wave_first(row0_reduce_sum(acc));       // nine non-zero lanes, all in row 0
      if (hxhy * (acc + ring) < tol_x || kc >= maxit) {
        result = snap;
        sweeps = kc;
        break;
      }
      ++kc;
      tdone += 2;
    }
  }
  if (mine) su[g] = result;
  wave_lds_fence<T>();
  return sweeps;
}

// The same solve with as few instructions per sweep as the arithmetic allows -- ONE wave issues an instruction every ~5-6
// cycles whatever it is, so lexgs_pipelined_5x5's ~270 instructions per sweep (per-lane time counters, ring / neighbour
// selects, both division variants, a three-deep history shifted on every update, a stop test per sweep) are what its ~1350
// cycles per sweep are made of.  Requirement: the ring of the iterate is zero (every coarsest problem below the top of a
// hierarchy is a correction equation with homogeneous boundary values).  Then
//  * the nine unknowns sit in a 4-wide lane grid (lane = 4 ci + cj, cj < 3) whose pad lanes hold 0.0: all four neighbours
//    are plain DPP row shifts by 1 and 4, bound_ctrl zeros standing in for the ring;
//  * the anti-diagonal parity classes update under two constant lane masks on alternating time steps (three masked
//    start-up steps);
//  * the sweep-k snapshot the stop test needs is, by a per-lane constant, the current value or the value at the start of
//    this / the previous sweep's pair of steps (two plain copies per sweep instead of a shifted history);
//  * all four 16-lane rows of the wave run the SAME solve, so -- after the first four sweeps of a warm start, which are
//    tested one by one -- row r evaluates the stop test of the r-th of four sweeps from its own copy of that snapshot: one residual / reduction
//    / compare per four sweeps.  The first sweep that meets the tolerance (or maxit) is the result; up to three sweeps
//    past it are speculative and dropped.
// Same expressions in the same order as lexgs_pipelined_5x5 (sums with an exact zero instead of a selected ring value), same
// sweep count; the nine squares of the stop test are added in another lane order.
template <typename T, bool VAR, bool EXACT>
__device__ int lexgs_5x5_zero_ring(T* __restrict__ su, const T* __restrict__ sf, T hx2, T hy2, T diag, T coeff, T omega, T one_m_omega,
                                   double hxhy, double tol_x, int maxit, int lane, const T* __restrict__ sa, T sigma) {
  constexpr int ny = 5;
  const T rhx2 = T(1) / hx2, rhy2 = T(1) / hy2, rdiag = T(1) / diag;
  double ring = 0.0;
  if (lane < 25) {
    const int i = lane / ny, j = lane - i * ny;
    if (i == 0 || i == 4 || j == 0 || j == 4) ring = (double)sf[lane] * (double)sf[lane];
  }
  ring = wave_first(wave_reduce_sum(ring));
  const int row = lane >> 4, cl = lane & 15;
  const int ci = cl >> 2, cj = cl & 3;
  const bool mine = cl < 11 && cj < 3;
  const int g = mine ? (ci + 1) * ny + cj + 1 : ny + 1, d = ci + cj + 2;
  const T fv = mine ? sf[g] : T(0);
  T uv = mine ? su[g] : T(0);
  T aip = T(1), aim = T(1), ajp = T(1), ajm = T(1), Dv = diag;
  if (VAR) {
    const T ac = sa[g];
    aip = T(0.5) * (ac + sa[g + ny]); aim = T(0.5) * (ac + sa[g - ny]); ajp = T(0.5) * (ac + sa[g + 1]); ajm = T(0.5) * (ac + sa[g - 1]);
    Dv = (EXACT ? (aip + aim) * rhx2 + (ajp + ajm) * rhy2 : (aip + aim) / hx2 + (ajp + ajm) / hy2) + sigma;
  }
  const T rDv = T(1) / Dv;
  const bool even = mine && !(d & 1), odd = mine && (d & 1);
  const bool lead1 = mine && (d == 3 || d == 4), lead2 = mine && d == 2;   // sweeps ahead of the sweep under test
  auto step = [&](bool upd) {
    const T up = dpp_row_move<0x114>(uv), dn = dpp_row_move<0x104>(uv);    // row_shr:4 / row_shl:4: lanes c -+ 4
    const T lf = dpp_row_move<0x111>(uv), rt = dpp_row_move<0x101>(uv);
    const T sx = VAR ? aip * dn + aim * up : dn + up, sy = VAR ? ajp * rt + ajm * lf : rt + lf;
    const T nb = EXACT ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2;
    const T num = fv + nb;
    const T un = VAR ? num * rDv : (EXACT ? num * rdiag : num / diag);
    const T nv = one_m_omega * uv + omega * un;
    uv = upd ? nv : uv;
  };
  // cell on anti-diagonal d takes its sweep-k value at time 2k + d: t = 4, 5, 6 start the pipeline, from t = 7 on every
  // odd (even) time step updates all odd (even) anti-diagonals.  p1 / p2: the values after time 2k + 4 / 2k + 2.
  step(mine && d == 2);
  T p1 = uv, p2 = uv;
  step(mine && d == 3);
  step(mine && (d == 2 || d == 4));
  int sweeps = maxit;
  T result = uv;
  // the stop test of ONE snapshot (every row evaluates the same one) / of four (row r: the r-th)
  auto sumsq = [&](T snap) {
    const T up = dpp_row_move<0x114>(snap), dn = dpp_row_move<0x104>(snap);
    const T lf = dpp_row_move<0x111>(snap), rt = dpp_row_move<0x101>(snap);
    const T sx = VAR ? aip * dn + aim * up : dn + up, sy = VAR ? ajp * rt + ajm * lf : rt + lf;
    const T rv = fv - coeff * ((EXACT ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2) - snap * (VAR ? Dv : diag));
    const double acc = mine ? (double)rv * (double)rv : 0.0;
    return row0_reduce_sum(acc);                   // per 16-lane row: the sum of its nine squares in the row's first lane
  };
  // A solve that starts from zero (the first visit of a coarsest problem) needs tens of sweeps: tested four at a time from
  // the start.  One that starts from an iterate (the second visit inside a W / F cycle) stops after a few: its first
  // kSingle sweeps are tested one by one, without speculative sweeps.
  constexpr int kSingle = 4;
  const int nsingle = (__ballot(uv != T(0)) == 0ull) ? 0 : kSingle;
  bool done = false;
  for (int k = 1; k <= nsingle; ++k) {
    p2 = p1;
    p1 = uv;
    step(odd);
    step(even);                                    // t = 2k + 6: the last cell has its sweep-k value
    const T snap = lead2 ? p2 : (lead1 ? p1 : uv);
    const double acc = wave_first(sumsq(snap));
    if (hxhy * (acc + ring) < tol_x || k >= maxit) {
      result = snap;
      sweeps = k;
      done = true;
      break;
    }
  }
  if (!done) {
    for (int k0 = nsingle + 1;; k0 += 4) {
      T s[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        p2 = p1;
        p1 = uv;
        step(odd);
        step(even);
        s[j] = lead2 ? p2 : (lead1 ? p1 : uv);
      }
      const double acc = sumsq(row == 0 ? s[0] : (row == 1 ? s[1] : (row == 2 ? s[2] : s[3])));
      const bool stop = cl == 0 && (hxhy * (acc + ring) < tol_x || k0 + row >= maxit);
      const unsigned long long m = __ballot(stop);
      if (m) {
        const int r = (__ffsll((long long)m) - 1) >> 4;   // the first of the four sweeps that stops
        result = r == 0 ? s[0] : (r == 1 ? s[1] : (r == 2 ? s[2] : s[3]));
        sweeps = k0 + r;
        break;
      }
    }
  }
  if (mine && row == 0) su[g] = result;
  wave_lds_fence<T>();
  return sweeps;
}

template <typename T, bool VAR = false>
__device__ int lexgs_pipelined(T* __restrict__ su, const T* __restrict__ sf, T* __restrict__ hist, int nx, int ny,
                               T hx2, T hy2, T diag, T coeff, T omega, T one_m_omega, bool exact, double hxhy,
                               double tol_x, int maxit, int lane, const T* __restrict__ sa = nullptr, T sigma = T(0)) {
  if (nx == 5 && ny == 5) {
    bool ring_nonzero = false;
    if (lane < 25) {
      const int i = lane / 5, j = lane - i * 5;
      ring_nonzero = (i == 0 || i == 4 || j == 0 || j == 4) && su[lane] != T(0);
    }
    if (__ballot(ring_nonzero) == 0ull)
      return exact ? lexgs_5x5_zero_ring<T, VAR, true>(su, sf, hx2, hy2, diag, coeff, omega, one_m_omega, hxhy, tol_x, maxit, lane, sa, sigma)
                   : lexgs_5x5_zero_ring<T, VAR, false>(su, sf, hx2, hy2, diag, coeff, omega, one_m_omega, hxhy, tol_x, maxit, lane, sa, sigma);
    return lexgs_pipelined_5x5<T, VAR>(su, sf, hx2, hy2, diag, coeff, omega, one_m_omega, exact, hxhy, tol_x, maxit, lane, sa, sigma);
  }
  // face means and diagonal of cell c (variable coefficient; coarse_lexgs_kernel's expressions)
  auto faces = [&](int c, T& aip, T& aim, T& ajp, T& ajm, T& Dv) {
    const T ac = sa[c];
    aip = T(0.5) * (ac + sa[c + ny]); aim = T(0.5) * (ac + sa[c - ny]); ajp = T(0.5) * (ac + sa[c + 1]); ajm = T(0.5) * (ac + sa[c - 1]);
    Dv = (exact ? (aip + aim) * rhx2_of(hx2) + (ajp + ajm) * rhx2_of(hy2) : (aip + aim) / hx2 + (ajp + ajm) / hy2) + sigma;
  };
  const int ncell = nx * ny, dmax = nx + ny - 4;
  const int H = (dmax - 1) / 2 + 2;
  const T rhx2 = T(1) / hx2, rhy2 = T(1) / hy2, rdiag = T(1) / diag;
  double ring = 0.0;
  for (int c = lane; c < ncell; c += 64) {
    const int i = c / ny, j = c - i * ny;
    if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) {
      ring += (double)sf[c] * (double)sf[c];
      for (int h = 0; h < H; ++h) hist[h * ncell + c] = su[c];
    }
  }
  ring = wave_reduce_sum(ring);
  ring = wave_first(ring);
  wave_lds_fence<T>();
  int sweeps = maxit;
  if (ncell <= 64) {
    // one cell per lane: everything that does not change with time lives in registers (a single wave issues
    // one instruction every few cycles, so the loop body is kept to the loads, ~10 flops and two stores)
    const int c = lane;
    const int ci = c / ny, cj = c - ci * ny;
    const bool interior = c < ncell && ci >= 1 && ci <= nx - 2 && cj >= 1 && cj <= ny - 2;
    const T fv = interior ? sf[c] : T(0);
    T uv = interior ? su[c] : T(0);
    T aip = T(1), aim = T(1), ajp = T(1), ajm = T(1), Dv = diag;
    if (VAR && interior) faces(c, aip, aim, ajp, ajm, Dv);
    const T rDv = T(1) / Dv;
    int tnext = 2 + ci + cj;            // time of this cell's next update: 2k + d with k = 1
    int knext = 1;
    T* hslot = hist + c;                // snapshot slot of sweep knext (slot index knext % H, advanced incrementally)
    int hs = 1 % H;
    hslot += hs * ncell;
    const T* snap = hist + (1 % H) * ncell;   // snapshot of the next sweep to complete
    int cs = 1 % H, kc = 1;
    int tdone = 2 + dmax;               // time at which sweep kc is complete
    for (int t = 4;; ++t) {
      if (interior && t == tnext && knext <= maxit) {
        const T sx = VAR ? aip * su[c + ny] + aim * su[c - ny] : su[c + ny] + su[c - ny];
        const T sy = VAR ? ajp * su[c + 1] + ajm * su[c - 1] : su[c + 1] + su[c - 1];
        const T nb = exact ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2;
        const T num = fv + nb;
        const T un = VAR ? num * rDv : (exact ? num * rdiag : num / diag);
        uv = one_m_omega * uv + omega * un;
        su[c] = uv;
        *hslot = uv;
        tnext += 2;
        ++knext;
        if (++hs == H) { hs = 0; hslot = hist + c; } else hslot += ncell;
      }
      wave_lds_fence<T>();
      if (t == tdone) {
        double acc = 0.0;
        if (interior) {
          const T sx = VAR ? aip * snap[c + ny] + aim * snap[c - ny] : snap[c + ny] + snap[c - ny];
          const T sy = VAR ? ajp * snap[c + 1] + ajm * snap[c - 1] : snap[c + 1] + snap[c - 1];
          const T rv = fv - coeff * ((exact ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2) - snap[c] * (VAR ? Dv : diag));
          acc = (double)rv * (double)rv;
        }
        acc = wave_reduce_sum(acc);
        acc = wave_first(acc);
        if (hxhy * (acc + ring) < tol_x || kc >= maxit) {
          if (c < ncell) su[c] = snap[c];
          sweeps = kc;
          break;
        }
        ++kc;
        tdone += 2;
        if (++cs == H) { cs = 0; snap = hist; } else snap += ncell;
      }
    }
    wave_lds_fence<T>();
    return sweeps;
  }
  for (int t = 4;; ++t) {
    for (int c = lane; c < ncell; c += 64) {
      const int i = c / ny, j = c - i * ny;
      if (i < 1 || i > nx - 2 || j < 1 || j > ny - 2) continue;
      const int td = t - (i + j);
      if (td & 1) continue;
      const int k = td >> 1;
      if (k < 1 || k > maxit) continue;
      T aip = T(1), aim = T(1), ajp = T(1), ajm = T(1), Dv = diag;
      if (VAR) faces(c, aip, aim, ajp, ajm, Dv);
      const T sx = VAR ? aip * su[c + ny] + aim * su[c - ny] : su[c + ny] + su[c - ny];
      const T sy = VAR ? ajp * su[c + 1] + ajm * su[c - 1] : su[c + 1] + su[c - 1];
      const T nb = exact ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2;
      const T num = sf[c] + nb;
      const T un = VAR ? num * (T(1) / Dv) : (exact ? num * rdiag : num / diag);
      const T v = one_m_omega * su[c] + omega * un;
      su[c] = v;
      hist[(k % H) * ncell + c] = v;
    }
    wave_lds_fence<T>();
    const int tc = t - dmax;
    if (tc >= 2 && !(tc & 1)) {
      const int kc = tc >> 1;
      const T* snap = hist + (kc % H) * ncell;
      double acc = 0.0;
      for (int c = lane; c < ncell; c += 64) {
        const int i = c / ny, j = c - i * ny;
        if (i < 1 || i > nx - 2 || j < 1 || j > ny - 2) continue;
        T aip = T(1), aim = T(1), ajp = T(1), ajm = T(1), Dv = diag;
        if (VAR) faces(c, aip, aim, ajp, ajm, Dv);
        const T sx = VAR ? aip * snap[c + ny] + aim * snap[c - ny] : snap[c + ny] + snap[c - ny];
        const T sy = VAR ? ajp * snap[c + 1] + ajm * snap[c - 1] : snap[c + 1] + snap[c - 1];
        const T rv = sf[c] - coeff * ((exact ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2) - snap[c] * (VAR ? Dv : diag));
        acc += (double)rv * (double)rv;
      }
      acc = wave_reduce_sum(acc);
      acc = wave_first(acc);
      if (hxhy * (acc + ring) < tol_x || kc >= maxit) {
        for (int c = lane; c < ncell; c += 64) su[c] = snap[c];
        sweeps = kc;
        break;
      }
    }
  }
  wave_lds_fence<T>();
  return sweeps;
}

// Small-grid variant (nx*ny <= kCoarseLdsCells): ONE wave, u and rhs live in LDS for the whole solve, the
// stop test is a wave64 shuffle reduction.  Same arithmetic and sweep order as coarse_lexgs_kernel; the L2
// round trip per anti-diagonal (the whole cost of a 5x5 solve) is gone.
constexpr int kCoarseLdsCells = 1089;   // up to 33 x 33
template <typename T>
__global__ __launch_bounds__(64) void coarse_lexgs_small_kernel(T* __restrict__ u, const T* __restrict__ rhs, int nx,
                                                                int ny, int ld, T hx2, T hy2, T omega, T one_m_omega,
                                                                T diag, T coeff, double hxhy, double tol, int maxit,
                                                                int* __restrict__ sweeps_out, int zero_init, int exact_recip, double tol_x) {
  // exact_recip: hx^2, hy^2 and the diagonal are powers of two, so x / c == x * (1/c) bit for bit and the
  // three IEEE divisions per cell (the reference divides, solvers/smoothers.py:165-170) become multiplications.
  const T rhx2 = T(1) / hx2, rhy2 = T(1) / hy2, rdiag = T(1) / diag;
  __shared__ T su[kCoarseLdsCells];
  __shared__ T sf[kCoarseLdsCells];
  __shared__ T hist[kPipeCells * kPipeSlots];
  const int lane = threadIdx.x;
  for (int idx = lane; idx < nx * ny; idx += 64) {
    const int i = idx / ny, j = idx - i * ny;
    su[idx] = zero_init ? T(0) : u[(size_t)i * ld + j];      // zero_init: the zero correction, ring included
    sf[idx] = rhs[(size_t)i * ld + j];
  }
  __syncthreads();
  if (nx * ny <= kPipeCells && (nx + ny - 5) / 2 + 2 <= kPipeSlots) {
    const int sw = lexgs_pipelined<T>(su, sf, hist, nx, ny, hx2, hy2, diag, coeff, omega, one_m_omega, exact_recip != 0,
                                      hxhy, tol_x, maxit, lane);
    __syncthreads();
    for (int idx = lane; idx < nx * ny; idx += 64) {
      const int i = idx / ny, j = idx - i * ny;
      if (i >= 1 && i < nx - 1 && j >= 1 && j < ny - 1) u[(size_t)i * ld + j] = su[idx];
    }
    if (lane == 0 && sweeps_out) *sweeps_out = sw;
    return;
  }
  int it = 0;
  for (it = 1; it <= maxit; ++it) {
    for (int sdiag = 2; sdiag <= nx + ny - 4; ++sdiag) {
      const int ilo = max(1, sdiag - (ny - 2)), ihi = min(nx - 2, sdiag - 1);
      for (int i = ilo + lane; i <= ihi; i += 64) {
        const int j = sdiag - i;
        T* p = su + i * ny + j;
        const T sx = p[ny] + p[-ny], sy = p[1] + p[-1];
        const T nb = exact_recip ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2;
        const T num = sf[i * ny + j] + nb;
        const T un = exact_recip ? num * rdiag : num / diag;
        p[0] = one_m_omega * p[0] + omega * un;
      }
      __syncthreads();
    }
    double acc = 0.0;
    for (int idx = lane; idx < nx * ny; idx += 64) {
      const int i = idx / ny, j = idx - i * ny;
      const T* p = su + idx;
      T rv = sf[idx];
      if (i >= 1 && i < nx - 1 && j >= 1 && j < ny - 1) {
        const T sx = p[ny] + p[-ny], sy = p[1] + p[-1];
        rv = rv - coeff * ((exact_recip ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2) - p[0] * diag);
      }
      acc += (double)rv * (double)rv;
    }
    acc = wave_reduce_sum(acc);
    acc = wave_first(acc);
    if (sqrt(hxhy * acc) < tol) break;
  }
  __syncthreads();
  for (int idx = lane; idx < nx * ny; idx += 64) {
    const int i = idx / ny, j = idx - i * ny;
    if (i >= 1 && i < nx - 1 && j >= 1 && j < ny - 1) u[(size_t)i * ld + j] = su[idx];
  }
  if (lane == 0 && sweeps_out) *sweeps_out = (it > maxit) ? maxit : it;
}

// ============================================================================================
// Variable-coefficient operator  A u = coeff * div(a grad u)  (coeff = -1: -div(a grad u) = f).
//   BASELINE config 5 names it; the reference has NO implementation, so this is our design and
//   its parity is "unpinned": pinned only by (i) a == 1 reproducing the constant-coefficient kernels bit for bit
//   on dyadic grids and (ii) second-order convergence on a manufactured solution (tests).
//   Discretisation: vertex values a[i][j]; face values by arithmetic mean, a(i+1/2,j) = 0.5 (a[i][j] + a[i+1][j]);
//     sx = a(i+1/2) u[i+1] + a(i-1/2) u[i-1],  sy likewise in j,  D = (a(i+1/2)+a(i-1/2))/hx^2 + (a(j+1/2)+a(j-1/2))/hy^2
//     A u   = coeff * ((sx/hx^2 + sy/hy^2) - u D)            residual r = f - A u (boundary r = f)
//     Jacobi: un = (f + sx/hx^2 + sy/hy^2) * (1 / D) ; u' = (1-w) u + w un ; red-black GS the same per colour, in place.
//       (round 3: times the reciprocal diagonal, rounded once -- it depends on a only, and the fused legs read it from a
//       field of its own (var_rdiag_kernel), so no sweep divides; until round 2 the sweeps divided by D)
//   Coarse operators: re-discretisation with a injected to the coarse vertices.
//   One kernel, four modes; u and a tiles (+1-cell halo) staged in LDS (37 KB -> 4 workgroups / CU); 4w / DoF.
// ============================================================================================
constexpr int kVarJacobi = 0, kVarRbgs = 1, kVarResidual = 2, kVarResidualNorm = 3;

template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void varcoef_kernel(const T* u_in, const T* __restrict__ a_in,
                                                         const T* __restrict__ rhs, T* out,   // out == u_in for red-black GS
                                                         double* __restrict__ partials, TileGeom g, T ihx2, T ihy2,
                                                         T omega, T one_m_omega, T coeff, int colour, int poff,
                                                         T sigma) {   // Helmholtz shift added to the diagonal (0: none)
  using S = TileShape<T>;
  __shared__ __attribute__((aligned(16))) T s[S::LDS_ELEMS];
  __shared__ __attribute__((aligned(16))) T sa[S::LDS_ELEMS];
  __shared__ double red[kBlock / 64];
  const int L = xcd_remap(blockIdx.x, g.ntiles);
  const int ti = L / g.tiles_j, tj = L - ti * g.tiles_j;
  const int i0 = g.i_org + ti * kTI, j0 = tj * S::TJ;
  const int cg = threadIdx.x % S::CG, rg = threadIdx.x / S::CG;
  const int gj0 = j0 + cg * S::N;
  const int lr = rg * S::RPT;

  Pack<T> f[S::RPT];
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const int gi = i0 + lr + k;
    f[k] = (gi < g.nx && gj0 < g.nyv) ? ldg(rhs + (size_t)gi * g.ld + gj0) : zero_pack<T>();
  }
  stage_tile<T>(u_in, s, i0, j0, g.nx, g.nyv, g.ld);
  stage_tile<T>(a_in, sa, i0, j0, g.nx, g.nyv, g.ld);
  __syncthreads();

  const int lc = S::N + cg * S::N;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const int r = lr + k + 1;                       // LDS row of the centre
    const Pack<T> up = *reinterpret_cast<const Pack<T>*>(s + (r - 1) * S::SJ + lc);
    const Pack<T> mid = *reinterpret_cast<const Pack<T>*>(s + r * S::SJ + lc);
    const Pack<T> dn = *reinterpret_cast<const Pack<T>*>(s + (r + 1) * S::SJ + lc);
    const T left = s[r * S::SJ + lc - 1], right = s[r * S::SJ + lc + S::N];
    const Pack<T> aup = *reinterpret_cast<const Pack<T>*>(sa + (r - 1) * S::SJ + lc);
    const Pack<T> amid = *reinterpret_cast<const Pack<T>*>(sa + r * S::SJ + lc);
    const Pack<T> adn = *reinterpret_cast<const Pack<T>*>(sa + (r + 1) * S::SJ + lc);
    const T aleft = sa[r * S::SJ + lc - 1], aright = sa[r * S::SJ + lc + S::N];
    const int gi = i0 + lr + k;
    const bool row_in = (gi >= 1) && (gi < g.nx - 1);
    Pack<T> o;
#pragma unroll
    for (int e = 0; e < S::N; ++e) {
      const T w = (e == 0) ? left : mid.v[e - 1];
      const T ea = (e == S::N - 1) ? right : mid.v[e + 1];
      const T aw = (e == 0) ? aleft : amid.v[e - 1];
      const T ae = (e == S::N - 1) ? aright : amid.v[e + 1];
      const T aip = T(0.5) * (amid.v[e] + adn.v[e]), aim = T(0.5) * (amid.v[e] + aup.v[e]);
      const T ajp = T(0.5) * (amid.v[e] + ae), ajm = T(0.5) * (amid.v[e] + aw);
      const T sx = aip * dn.v[e] + aim * up.v[e];
      const T sy = ajp * ea + ajm * w;
      const T D0 = (aip + aim) * ihx2 + (ajp + ajm) * ihy2;
      const T D = (sigma != T(0)) ? D0 + sigma : D0;
      const int gj = gj0 + e;
      const bool interior = row_in && gj >= 1 && gj < g.ny - 1;
      if (MODE == kVarJacobi || MODE == kVarRbgs) {
        const T nb = ihx2 * sx + ihy2 * sy;
        const T un = (f[k].v[e] + nb) * (T(1) / D);      // times the reciprocal diagonal (the fused legs read it from the level's rdiag field)
        const T res = one_m_omega * mid.v[e] + omega * un;
        const bool mine = (MODE == kVarJacobi) || (((gi + gj + poff) & 1) == colour);
        o.v[e] = (interior && mine) ? res : mid.v[e];
      } else {
        const T au = coeff * ((sx * ihx2 + sy * ihy2) - mid.v[e] * D);
        const T rv = interior ? (f[k].v[e] - au) : f[k].v[e];
        o.v[e] = rv;
        if (MODE == kVarResidualNorm && gi < g.nx && gj < g.ny) acc += (double)rv * (double)rv;
      }
    }
    if (MODE != kVarResidualNorm && gi < g.nx && gj0 < g.nyv) stg(out + (size_t)gi * g.ld + gj0, o);
  }
  if (MODE == kVarResidualNorm) {
    const double t = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// Reciprocal diagonal of the variable-coefficient operator, one value per cell: rd = 1 / D with D exactly as the residual
// stages form it -- (aip + aim) ihx2 + (ajp + ajm) ihy2 [+ sigma], face means 0.5 (a_c + a_nb) -- rounded ONCE in the level's
// dtype; 0 on boundary cells.  Recomputed when the coefficient or the shift changes; every sweep multiplies by it.
template <typename T>
__global__ __launch_bounds__(kBlock) void var_rdiag_kernel(const T* __restrict__ a, T* __restrict__ rd, int nx, int ny, int ld, T ihx2,
                                                           T ihy2, T sigma) {
  const long long total = (long long)nx * ny;
  for (long long v = (long long)blockIdx.x * kBlock + threadIdx.x; v < total; v += (long long)gridDim.x * kBlock) {
    const int i = (int)(v / ny), j = (int)(v - (long long)i * ny);
    T out = T(0);
    if (i >= 1 && i < nx - 1 && j >= 1 && j < ny - 1) {
      const T* q = a + (size_t)i * ld + j;
      const T aip = T(0.5) * (q[0] + q[ld]), aim = T(0.5) * (q[0] + q[-ld]), ajp = T(0.5) * (q[0] + q[1]), ajm = T(0.5) * (q[0] + q[-1]);
      const T D0 = (aip + aim) * ihx2 + (ajp + ajm) * ihy2;
      out = T(1) / ((sigma != T(0)) ? D0 + sigma : D0);
    }
    rd[(size_t)i * ld + j] = out;
  }
}

// full injection fine -> coarse with a stride (coefficient field of the re-discretised coarse operators: level l takes
// every 2^l-th vertex value of the fine field, cast to the level's precision)
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void inject_kernel(const TIN* __restrict__ fine, TOUT* __restrict__ coarse, int ldf,
                                                        int nxc, int nyc, int ldc, int stride) {
  const long long total = (long long)nxc * nyc;
  for (long long v = (long long)blockIdx.x * kBlock + threadIdx.x; v < total; v += (long long)gridDim.x * kBlock) {
    const int ic = (int)(v / nyc), jc = (int)(v - (long long)ic * nyc);
    coarse[(size_t)ic * ldc + jc] = (TOUT)fine[(size_t)(stride * ic) * ldf + (size_t)stride * jc];
  }
}

// ============================================================================================
// Coarse tail: every level with <= ~65^2 cells, down to and including the coarsest-grid solve, runs in ONE
// workgroup with all of its fields resident in LDS (137 KB for 65^2..5^2 in fp64).  A V-cycle visits those
// levels with ~12 launches of ~5 us each (launch-latency floor); a W-cycle visits the coarsest one 2^(L-1)
// times.  Here the whole sub-cycle is one launch that interprets a host-built schedule of
// {down leg, coarsest solve, up leg} steps.  Arithmetic per cell is identical to the per-operator kernels.
// ============================================================================================
#ifndef MG_TAIL_BLOCK
#define MG_TAIL_BLOCK 1024
#endif
constexpr int kTailBlock = MG_TAIL_BLOCK;
constexpr int kTailMaxLevels = 6;
constexpr int kTailDown = 0, kTailSolve = 1, kTailUp = 2;

struct TailLevel {
  int nx, ny;
  int off;                     // byte offset of this level's arrays in the LDS pool
  double ihx2, ihy2, invD, diag, hx2, hy2, hxhy;
  int use_div;
  int exact_recip;             // hx^2, hy^2 and the diagonal are all powers of two
};
struct TailArgs {
  int nlev, nops, pre, post, ld_top, maxit;
  int smoother;                // kSmJacobi / kSmRbgs
  int colour_offset;
  double omega, coeff, tol;
  double tol_x;                // min{x : sqrt(x) >= tol}: the coarsest stop test without the square root (sqrt_threshold)
  double sigma;                // Helmholtz shift of the variable-coefficient diagonal (constant path: folded into diag)
  TailLevel lv[kTailMaxLevels];
  const void* a_lv[kTailMaxLevels];   // VAR: the coefficient field of every tail level in HBM (dtype of that level)
  int a_ld[kTailMaxLevels];
  int direct;                  // 1: the 5 x 5 coarsest system (nine unknowns, zero ring) is solved by u = minv f instead of the
                               // reference's Gauss-Seidel iteration to coarse_tol (mg_config.coarse_direct; not bit-identical)
  double minv[81];             // inverse of the 9 x 9 coarsest matrix, row-major (host, long double elimination)
  const double* minv_dev;      // any other coarsest grid with n = (nx - 2) (ny - 2) <= 64 unknowns (the 9 x 5 of a 2:1 domain, ...): its
  int minv_n;                  // n x n inverse in device memory, row-major, unknown (i, j) at (i - 1) (ny - 2) + (j - 1)
};

// variable coefficient: the relaxed value of cell idx from the vertex values `A` (varcoef_kernel's expressions)
template <typename T>
__device__ __forceinline__ T tail_var_un(const T* __restrict__ src, const T* __restrict__ A, T fv, int idx, int ny, T ihx2, T ihy2,
                                         T sigma) {
  const T ac = A[idx];
  const T aip = T(0.5) * (ac + A[idx + ny]), aim = T(0.5) * (ac + A[idx - ny]);
  const T ajp = T(0.5) * (ac + A[idx + 1]), ajm = T(0.5) * (ac + A[idx - 1]);
  const T sx = aip * src[idx + ny] + aim * src[idx - ny], sy = ajp * src[idx + 1] + ajm * src[idx - 1];
  const T D0 = (aip + aim) * ihx2 + (ajp + ajm) * ihy2;
  return (fv + (ihx2 * sx + ihy2 * sy)) * (T(1) / ((sigma != T(0)) ? D0 + sigma : D0));
}

template <typename T, bool VAR>
__device__ __forceinline__ void tail_sweep(const T* __restrict__ src, T* __restrict__ dst, const T* __restrict__ f,
                                           const TailLevel& L, T omega, T one_m_omega, const T* __restrict__ A, T sigma) {
  const T ihx2 = (T)L.ihx2, ihy2 = (T)L.ihy2, invD = (T)L.invD, D = (T)L.diag;
  const int ny = L.ny, ni = L.nx - 2, nj = ny - 2;
  for (int c = threadIdx.x; c < ni * nj; c += kTailBlock) {
    const int i = 1 + c / nj, j = 1 + c % nj, idx = i * ny + j;
    T un;
    if (VAR) {
      un = tail_var_un<T>(src, A, f[idx], idx, ny, ihx2, ihy2, sigma);
    } else {
      const T nb = ihx2 * (src[idx + ny] + src[idx - ny]) + ihy2 * (src[idx + 1] + src[idx - 1]);
      un = L.use_div ? (f[idx] + nb) / D : (f[idx] + nb) * invD;
    }
    dst[idx] = one_m_omega * src[idx] + omega * un;
  }
}

// one colour pass of red-black GS, in place (solvers/smoothers.py:183-205)
template <typename T, bool VAR>
__device__ __forceinline__ void tail_rb_pass(T* __restrict__ u, const T* __restrict__ f, const TailLevel& L, T omega,
                                             T one_m_omega, int colour, int poff, const T* __restrict__ A, T sigma) {
  const T ihx2 = (T)L.ihx2, ihy2 = (T)L.ihy2, invD = (T)L.invD, D = (T)L.diag;
  const int ny = L.ny, ni = L.nx - 2, nj = ny - 2;
  for (int c = threadIdx.x; c < ni * nj; c += kTailBlock) {
    const int i = 1 + c / nj, j = 1 + c % nj, idx = i * ny + j;
    if (((i + j + poff) & 1) != colour) continue;
    T un;
    if (VAR) {
      un = tail_var_un<T>(u, A, f[idx], idx, ny, ihx2, ihy2, sigma);
    } else {
      const T nb = ihx2 * (u[idx + ny] + u[idx - ny]) + ihy2 * (u[idx + 1] + u[idx - 1]);
      un = L.use_div ? (f[idx] + nb) / D : (f[idx] + nb) * invD;
    }
    u[idx] = one_m_omega * u[idx] + omega * un;
  }
}

// Timing-only switches (results are WRONG by construction with NO_LOAD / NO_COMPUTE / NO_STORE) exist in measurement
// builds only: -DMG_EXPERIMENTS, built to a file of its own (tools/README.md).  The shipped library cannot carry them.
#if !defined(MG_EXPERIMENTS) && (defined(MG_EXP_NO_LOAD) || defined(MG_EXP_NO_COMPUTE) || defined(MG_EXP_NO_STORE) || defined(MG_EXP_TAIL_TRACE))
#error "MG_EXP_* switches need -DMG_EXPERIMENTS (a measurement build, never the shipped library)"
#endif
#ifndef MG_EXP_TAIL_TRACE
#define MG_EXP_TAIL_TRACE 0
#endif
#if MG_EXP_TAIL_TRACE
__device__ long long g_tail_trace[64];     // timing experiment: s_memtime at entry, after the prologue, after every op, at exit
#define TAIL_STAMP(k) do { if (threadIdx.x == 0 && (k) < 64) g_tail_trace[k] = clock64(); } while (0)
#else
#define TAIL_STAMP(k) do { } while (0)
#endif
template <typename T, typename TCO, typename TC, bool VAR = false>
__global__ __launch_bounds__(kTailBlock) void coarse_tail_kernel(const T* __restrict__ rhs_top, T* __restrict__ u_top,
                                                                 const int* __restrict__ ops, TailArgs a, int zero_top,
                                                                 int* __restrict__ sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pool[];
  TAIL_STAMP(0);
  const int last = a.nlev - 1;
  const T omega = (T)a.omega, one_m_omega = (T)(1.0 - a.omega), coeff = (T)a.coeff, sigma = (T)a.sigma;
  // per level: two iterate buffers (ping-pong), the rhs [and the coefficient: VAR]; the last level holds {u, rhs [, a]} in TCO
  auto Ubuf = [&](int l, int which) -> T* { return reinterpret_cast<T*>(pool + a.lv[l].off) + (size_t)which * a.lv[l].nx * a.lv[l].ny; };
  auto Fbuf = [&](int l) -> T* { return reinterpret_cast<T*>(pool + a.lv[l].off) + (size_t)2 * a.lv[l].nx * a.lv[l].ny; };
  auto Abuf = [&](int l) -> T* { return VAR ? reinterpret_cast<T*>(pool + a.lv[l].off) + (size_t)3 * a.lv[l].nx * a.lv[l].ny : nullptr; };
  TCO* const Ulast = reinterpret_cast<TCO*>(pool + a.lv[last].off);
  TCO* const Flast = Ulast + a.lv[last].nx * a.lv[last].ny;
  TCO* const Alast = Flast + a.lv[last].nx * a.lv[last].ny;                 // VAR only
  TCO* const Hist = VAR ? Alast + a.lv[last].nx * a.lv[last].ny : Alast;    // snapshot ring of the pipelined coarsest solve

  {   // zero the pool (rings of every iterate buffer stay zero for the whole launch), then load the top level
    const int quads = (a.lv[last].off + (VAR ? 3 : 2) * a.lv[last].nx * a.lv[last].ny * (int)sizeof(TCO) + 15) / 16;
    int4* w = reinterpret_cast<int4*>(pool);
    for (int c = threadIdx.x; c < quads; c += kTailBlock) w[c] = make_int4(0, 0, 0, 0);
  }
  __syncthreads();
  {
    const TailLevel& L0 = a.lv[0];
    T* f0 = Fbuf(0);
    T* u0 = Ubuf(0, 0);
    for (int c = threadIdx.x; c < L0.nx * L0.ny; c += kTailBlock) {
      const int i = c / L0.ny, j = c - i * L0.ny;
      f0[c] = rhs_top[(size_t)i * a.ld_top + j];
      if (!zero_top && i >= 1 && i < L0.nx - 1 && j >= 1 && j < L0.ny - 1) u0[c] = u_top[(size_t)i * a.ld_top + j];
    }
    if (VAR) {      // the coefficient of every tail level (a few KB, L2-resident between visits)
      for (int l = 0; l <= last; ++l) {
        const TailLevel& Ll = a.lv[l];
        for (int c = threadIdx.x; c < Ll.nx * Ll.ny; c += kTailBlock) {
          const int i = c / Ll.ny, j = c - i * Ll.ny;
          if (l == last) Alast[c] = reinterpret_cast<const TCO*>(a.a_lv[l])[(size_t)i * a.a_ld[l] + j];
          else Abuf(l)[c] = reinterpret_cast<const T*>(a.a_lv[l])[(size_t)i * a.a_ld[l] + j];
        }
      }
    }
  }
  __syncthreads();

  unsigned cur = 0;   // bit l: which iterate buffer of level l is current
  TAIL_STAMP(1);
  for (int ip = 0; ip < a.nops; ++ip) {
    if (ip > 0) TAIL_STAMP(1 + ip);
    const int op = ops[ip];
    const int code = op & 0xff, l = (op >> 8) & 0xff, zflag = (op >> 16) & 0xff;
    const bool zero = (zflag == 2) ? (zero_top != 0) : (zflag != 0);
    if (code == kTailDown) {
      const TailLevel& L = a.lv[l];
      const int ny = L.ny, ni = L.nx - 2, nj = ny - 2;
      if (zero && l > 0) {          // the zero correction (l == 0 was loaded / zeroed at entry)
        T* u = Ubuf(l, (cur >> l) & 1);
        for (int c = threadIdx.x; c < ni * nj; c += kTailBlock) u[(1 + c / nj) * ny + 1 + c % nj] = T(0);
        __syncthreads();
      }
      for (int s = 0; s < a.pre; ++s) {
        if (a.smoother == kSmRbgs) {
          for (int colour = 0; colour < 2; ++colour) {
            tail_rb_pass<T, VAR>(Ubuf(l, (cur >> l) & 1), Fbuf(l), L, omega, one_m_omega, colour, a.colour_offset, Abuf(l), sigma);
            __syncthreads();
          }
        } else {
          tail_sweep<T, VAR>(Ubuf(l, (cur >> l) & 1), Ubuf(l, ((cur >> l) & 1) ^ 1), Fbuf(l), L, omega, one_m_omega, Abuf(l), sigma);
          cur ^= (1u << l);
          __syncthreads();
        }
      }
      // residual of interior cells into the non-current buffer (its ring is never touched)
      const T* u = Ubuf(l, (cur >> l) & 1);
      T* r = Ubuf(l, ((cur >> l) & 1) ^ 1);
      const T* f = Fbuf(l);
      {
        const T ihx2 = (T)L.ihx2, ihy2 = (T)L.ihy2, D = (T)L.diag;
        const T* A = Abuf(l);
        for (int c = threadIdx.x; c < ni * nj; c += kTailBlock) {
          const int idx = (1 + c / nj) * ny + 1 + c % nj;
          T au;
          if (VAR) {
            const T ac = A[idx];
            const T aip = T(0.5) * (ac + A[idx + ny]), aim = T(0.5) * (ac + A[idx - ny]);
            const T ajp = T(0.5) * (ac + A[idx + 1]), ajm = T(0.5) * (ac + A[idx - 1]);
            const T sx = aip * u[idx + ny] + aim * u[idx - ny], sy = ajp * u[idx + 1] + ajm * u[idx - 1];
            const T D0 = (aip + aim) * ihx2 + (ajp + ajm) * ihy2;
            au = coeff * ((sx * ihx2 + sy * ihy2) - u[idx] * ((sigma != T(0)) ? D0 + sigma : D0));
          } else {
            au = coeff * (((u[idx + ny] + u[idx - ny]) * ihx2 + (u[idx + 1] + u[idx - 1]) * ihy2) - u[idx] * D);
          }
          r[idx] = f[idx] - au;
        }
      }
      __syncthreads();
      // full weighting into the next level's rhs; coarse boundary = injection of r = f
      const TailLevel& Lc = a.lv[l + 1];
      for (int c = threadIdx.x; c < Lc.nx * Lc.ny; c += kTailBlock) {
        const int ic = c / Lc.ny, jc = c - ic * Lc.ny;
        const int fidx = (2 * ic) * ny + 2 * jc;
        T val;
        if (ic == 0 || ic == Lc.nx - 1 || jc == 0 || jc == Lc.ny - 1) {
          val = f[fidx];
        } else {
          const T* p = r + fidx;
          const T corners = ((p[-ny - 1] + p[-ny + 1]) + p[ny - 1]) + p[ny + 1];
          const T edges = ((p[-ny] + p[ny]) + p[-1]) + p[1];
          val = (T(1.0 / 16.0) * corners + T(1.0 / 8.0) * edges) + T(1.0 / 4.0) * p[0];
        }
        if (l + 1 == last) Flast[c] = (TCO)val; else Fbuf(l + 1)[c] = val;
      }
      __syncthreads();
    } else if (code == kTailSolve) {
      // coarsest level: lexicographic GS by ONE wave (anti-diagonal order, wave-level sync only)
      if (threadIdx.x < 64) {
        const TailLevel& L = a.lv[last];
        const int nx = L.nx, ny = L.ny, lane = threadIdx.x;
        volatile TCO* su = Ulast;
        const volatile TCO* sf = Flast;
        const TCO hx2 = (TCO)L.hx2, hy2 = (TCO)L.hy2, diag = (TCO)L.diag, cf = (TCO)a.coeff;
        const TCO rhx2 = TCO(1) / hx2, rhy2 = TCO(1) / hy2, rdiag = TCO(1) / diag;
        const bool exact = L.exact_recip != 0;      // powers of two: x / c == x * (1/c) bit for bit
        if (zero) {
          for (int c = lane; c < nx * ny; c += 64) su[c] = TCO(0);
          __builtin_amdgcn_wave_barrier();
        }
        if (a.direct && nx == 5 && ny == 5) {
          // nine unknowns in lanes 0..8: u_i = sum_j minv[i][j] f_j, f_j broadcast from lane j; a fixed summation order
          const int li = lane < 9 ? lane : 0;
          const int g = (li / 3 + 1) * 5 + (li % 3) + 1;
          const double fv = (double)sf[g];
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < 9; ++j) {
            const double fj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fv), j), __builtin_amdgcn_readlane(__double2loint(fv), j));
            acc += a.minv[li * 9 + j] * fj;
          }
          if (lane < 9) su[g] = (TCO)acc;
          __builtin_amdgcn_wave_barrier();
          if (lane == 0 && sweeps_out) *sweeps_out = 0;
        } else if (a.direct && a.minv_dev && a.minv_n == (nx - 2) * (ny - 2)) {
          // the same for any coarsest grid of at most 64 unknowns (one per lane): row `lane` of the inverse streams from memory
          const int n = a.minv_n, my = ny - 2;
          const int li = lane < n ? lane : 0;
          const int g = (li / my + 1) * ny + (li % my) + 1;
          const double fv = (double)sf[g];
          const double* __restrict__ row = a.minv_dev + (size_t)li * n;
          double acc = 0.0;
          for (int j = 0; j < n; ++j) {
            const double fj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(fv), j), __builtin_amdgcn_readlane(__double2loint(fv), j));
            acc += row[j] * fj;
          }
          if (lane < n) su[g] = (TCO)acc;
          __builtin_amdgcn_wave_barrier();
          if (lane == 0 && sweeps_out) *sweeps_out = 0;
        } else if (nx * ny <= kPipeCells && (nx + ny - 5) / 2 + 2 <= kPipeSlots) {
          // kPipeSlots snapshots behind the last level's arrays
          const int sw = lexgs_pipelined<TCO, VAR>(Ulast, Flast, Hist, nx, ny, hx2, hy2, diag, cf, TCO(1), TCO(0), exact, L.hxhy,
                                                   a.tol_x, a.maxit, lane, VAR ? Alast : nullptr, (TCO)a.sigma);
          if (lane == 0 && sweeps_out) *sweeps_out = sw;
        } else {
        int it = 0;
        for (it = 1; it <= a.maxit; ++it) {
          for (int sdiag = 2; sdiag <= nx + ny - 4; ++sdiag) {
            const int ilo = max(1, sdiag - (ny - 2)), ihi = min(nx - 2, sdiag - 1);
            for (int i = ilo + lane; i <= ihi; i += 64) {
              const int idx = i * ny + (sdiag - i);
              TCO aip = TCO(1), aim = TCO(1), ajp = TCO(1), ajm = TCO(1), Dv = diag;
              if (VAR) {
                const TCO ac = Alast[idx];
                aip = TCO(0.5) * (ac + Alast[idx + ny]); aim = TCO(0.5) * (ac + Alast[idx - ny]);
                ajp = TCO(0.5) * (ac + Alast[idx + 1]); ajm = TCO(0.5) * (ac + Alast[idx - 1]);
                Dv = (exact ? (aip + aim) * rhx2 + (ajp + ajm) * rhy2 : (aip + aim) / hx2 + (ajp + ajm) / hy2) + (TCO)a.sigma;
              }
              const TCO sx = VAR ? aip * su[idx + ny] + aim * su[idx - ny] : su[idx + ny] + su[idx - ny];
              const TCO sy = VAR ? ajp * su[idx + 1] + ajm * su[idx - 1] : su[idx + 1] + su[idx - 1];
              const TCO nb = exact ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2;
              const TCO num = sf[idx] + nb;
              const TCO un = VAR ? num * (TCO(1) / Dv) : (exact ? num * rdiag : num / diag);
              su[idx] = TCO(0) * su[idx] + TCO(1) * un;      // omega = 1: (1-w)*u + w*un, as the reference evaluates it
            }
            __builtin_amdgcn_wave_barrier();
          }
          double acc = 0.0;
          for (int c = lane; c < nx * ny; c += 64) {
            const int i = c / ny, j = c - i * ny;
            TCO rv = sf[c];
            if (i >= 1 && i < nx - 1 && j >= 1 && j < ny - 1) {
              TCO aip = TCO(1), aim = TCO(1), ajp = TCO(1), ajm = TCO(1), Dv = diag;
              if (VAR) {
                const TCO ac = Alast[c];
                aip = TCO(0.5) * (ac + Alast[c + ny]); aim = TCO(0.5) * (ac + Alast[c - ny]);
                ajp = TCO(0.5) * (ac + Alast[c + 1]); ajm = TCO(0.5) * (ac + Alast[c - 1]);
                Dv = (exact ? (aip + aim) * rhx2 + (ajp + ajm) * rhy2 : (aip + aim) / hx2 + (ajp + ajm) / hy2) + (TCO)a.sigma;
              }
              const TCO sx = VAR ? aip * su[c + ny] + aim * su[c - ny] : su[c + ny] + su[c - ny];
              const TCO sy = VAR ? ajp * su[c + 1] + ajm * su[c - 1] : su[c + 1] + su[c - 1];
              rv = rv - cf * ((exact ? sx * rhx2 + sy * rhy2 : sx / hx2 + sy / hy2) - su[c] * Dv);
            }
            acc += (double)rv * (double)rv;
          }
          acc = wave_reduce_sum(acc);
          acc = wave_first(acc);
          if (sqrt(L.hxhy * acc) < a.tol) break;
        }
        if (lane == 0 && sweeps_out) *sweeps_out = (it > a.maxit) ? a.maxit : it;
        }
      }
      __syncthreads();
    } else {   // kTailUp: u_l += P u_{l+1}, then the post sweeps
      using TS = typename std::conditional<(sizeof(TC) > sizeof(T)), TC, T>::type;
      const TailLevel& L = a.lv[l];
      const TailLevel& Lc = a.lv[l + 1];
      const int nx = L.nx, ny = L.ny, nyc = Lc.ny;
      T* u = Ubuf(l, (cur >> l) & 1);
      const T* ec = (l + 1 == last) ? nullptr : Ubuf(l + 1, (cur >> (l + 1)) & 1);
      for (int c = threadIdx.x; c < nx * ny; c += kTailBlock) {
        const int i = c / ny, j = c - i * ny;
        const int ic = i >> 1, jc = j >> 1;
        const bool iodd = i & 1, jodd = j & 1;
        const int b0 = ic * nyc + jc, b1 = b0 + (iodd ? nyc : 0);
        TC e00, e01, e10, e11;
        if (l + 1 == last) { e00 = (TC)Ulast[b0]; e01 = jodd ? (TC)Ulast[b0 + 1] : e00; e10 = (TC)Ulast[b1]; e11 = jodd ? (TC)Ulast[b1 + 1] : e10; }
        else { e00 = (TC)ec[b0]; e01 = jodd ? (TC)ec[b0 + 1] : e00; e10 = (TC)ec[b1]; e11 = jodd ? (TC)ec[b1 + 1] : e10; }
        TC val;
        if (!iodd && !jodd) val = e00;
        else if (iodd && !jodd) val = (j == ny - 1) ? TC(0) : TC(0.5) * (e00 + e10);
        else if (!iodd && jodd) val = (i == nx - 1) ? TC(0) : TC(0.5) * (e00 + e01);
        else val = TC(0.25) * (((e00 + e01) + e10) + e11);
        u[c] = (T)((TS)u[c] + (TS)val);
      }
      __syncthreads();
      for (int s = 0; s < a.post; ++s) {
        if (a.smoother == kSmRbgs) {
          for (int colour = 0; colour < 2; ++colour) {
            tail_rb_pass<T, VAR>(Ubuf(l, (cur >> l) & 1), Fbuf(l), L, omega, one_m_omega, colour, a.colour_offset, Abuf(l), sigma);
            __syncthreads();
          }
        } else {
          tail_sweep<T, VAR>(Ubuf(l, (cur >> l) & 1), Ubuf(l, ((cur >> l) & 1) ^ 1), Fbuf(l), L, omega, one_m_omega, Abuf(l), sigma);
          cur ^= (1u << l);
          __syncthreads();
        }
      }
    }
  }
  TAIL_STAMP(1 + a.nops);
  {   // the correction of the top level back to HBM (interior cells; the ring is zero and stays zero there)
    const TailLevel& L0 = a.lv[0];
    const T* u0 = Ubuf(0, cur & 1);
    for (int c = threadIdx.x; c < L0.nx * L0.ny; c += kTailBlock) {
      const int i = c / L0.ny, j = c - i * L0.ny;
      if (i >= 1 && i < L0.nx - 1 && j >= 1 && j < L0.ny - 1) u_top[(size_t)i * a.ld_top + j] = u0[c];
    }
  }
  TAIL_STAMP(2 + a.nops);
}

// ============================================================================================
// Fused, temporally blocked smoothing stages (weighted Jacobi).
//
// One launch does a whole "leg" of the V-cycle on one level, so the fine arrays cross HBM once per
// leg instead of once per operator:
//   down leg  (POST = kPostRestrict):  nsweep Jacobi sweeps -> residual -> full-weighting restriction
//                                      reads u, rhs; writes u', coarse rhs (interior)      3.25 w / DoF
//   up leg    (PROLONG, POST = kPostNorm / kPostNone): u += P e -> nsweep sweeps [-> sum r^2]
//                                      reads u, rhs, e; writes u'                           3.25 w / DoF
//   plain     (POST = kPostNone, !PROLONG): nsweep sweeps                                    3 w / DoF
// against (nu+1)*3w + 1.25w and 2.25w + nu*3w (+2w for the norm) when every operator is its own launch.
//
// A workgroup (512 threads) owns a TI x TJ tile and stages the tile plus a HALO-cell ring of u in LDS
// (two buffers, ping-pong); every stage is evaluated on the ring too (redundantly with the neighbour
// tiles), shrinking by one cell per stage, so the tile itself ends up exact.  Each thread keeps the rhs
// of its fixed 3-row x 16-byte strip in registers.  Arithmetic per cell is the same sequence as in the
// single-operator kernels, hence bit-identical results.
//
// Invariants the driver maintains: boundary cells of u never change and both ping-pong buffers carry
// them; coarse-level rhs boundary cells (injection of r = f) are written once when the rhs is set.
// ============================================================================================
#ifndef MG_FUSED_BLOCK
#define MG_FUSED_BLOCK 512
#endif
constexpr int kFusedBlock = MG_FUSED_BLOCK;      // threads of a fused-leg workgroup (512: 3 workgroups / CU; 1024 measured slower, profiles/README.md)
#ifndef MG_FUSED_TI
#define MG_FUSED_TI 32
#endif
// Profiling experiments (timing-only builds, results are wrong): MG_EXP_NO_LOAD drops the global loads of the fused
// legs, MG_EXP_NO_COMPUTE their LDS stages, MG_EXP_NO_STORE their global stores -- how much of a leg is memory phase,
// how much compute, how much overlaps (profiles/README.md, round-1 experiment table).
#ifndef MG_EXP_NO_LOAD
#define MG_EXP_NO_LOAD 0
#endif
#ifndef MG_EXP_NO_COMPUTE
#define MG_EXP_NO_COMPUTE 0
#endif
#ifndef MG_EXP_NO_STORE
#define MG_EXP_NO_STORE 0
#endif
constexpr int kFusedTI = MG_FUSED_TI;      // tile rows of the fused legs on large levels (even: coarse rows sit on every other tile row)
constexpr int kFusedTISmall = 16;          // ... on levels of <= ~1025^2 cells: twice the workgroups, half the critical path of a launch
                                           // that is latency-bound anyway (4.4 vs 5.7 us per leg at 129^2-513^2)
constexpr int kFusedTITiny = 8;            // ... on levels of <= ~520^2 cells (513^2 ... 65^2): a launch there is one chain of load -> stages ->
                                           // store per workgroup; shorter tiles shorten the chain (constant-coefficient legs)
constexpr int kPostNone = 0, kPostRestrict = 1, kPostNorm = 2;

template <typename T, int HALO, int TI = kFusedTI> struct FusedShape {
  static constexpr int N = VecW<T>::N;
  static constexpr int TJ = kTileRowBytes / (int)sizeof(T);
  static constexpr int HV = (HALO + N - 1) / N;               // halo vectors per side
  static constexpr int RI = TI + 2 * HALO;                         // region rows
  static constexpr int RJ = TJ + 2 * HV * N;                  // region cols
  static constexpr int VPR = RJ / N;                          // vectors per region row
  static constexpr int RG = kFusedBlock / VPR;                // row groups
  static constexpr int RPT = (RI + RG - 1) / RG;              // rows per thread
  static constexpr int ELEMS = RI * RJ;
};

struct FusedArgs {
  int nx, ny, ld, nyv;          // fine level
  int tiles_j, ntiles;
  int nsweep;
  int nsweep2;                  // spanning leg (rb_span_kernel): pre sweeps of the following cycle (nsweep: post sweeps of this one)
  int band;                     // spanning leg: tile rows per band (tiles are numbered down the columns of a band: see the kernel)
  int colour_offset;            // parity of the global index of local cell (0,0) (red-black colouring)
  int use_div;                  // 1: divide by the diagonal (1/D not exact)
  int nxc, nyc, ldc;            // coarse level (restriction target / prolongation source)
  int ci_off, cj_off;           // coarse (ic, jc) <-> fine (2 (ic - ci_off), 2 (jc - cj_off)); 0 for whole grids
  int sides;                    // physical-boundary edges of this array (far-edge rule of the prolongation)
  int ni_lo, ni_hi, nj_lo, nj_hi;   // cells counted by the norm stage (whole grid: the interior)
  // tile selection (overlap of a halo exchange with the tiles that do not need it): 0 all tiles, 1 only tiles whose
  // staged region lies inside [in_i_lo, in_i_hi) x [in_j_lo, in_j_hi), 2 only the others
  int select, in_i_lo, in_i_hi, in_j_lo, in_j_hi;
  int exp_flags;                // bit 3 (8): non-temporal stores of the output tile, bit 4 (16): non-temporal loads of u, bit 5 (32) of rhs
                                // (set by the launcher for arrays that cannot stay in the 256 MiB Infinity Cache between two legs);
                                // timing experiments (MG_EXP_FLAGS): 1 no XCD remap, 2 column-major tile order, 4 no interior body
};

// VAR: the variable-coefficient operator A = coeff * div(a grad .) (see varcoef_kernel above for the discretisation and
// its association order, which the stages below repeat).  A thread owns the same RPT x N cells in every stage, so the
// face coefficients of its cells -- (RPT + 1) x N vertical, RPT x (N + 1) horizontal arithmetic means of the vertex values
// -- are formed ONCE from an LDS-staged tile of `a` and then live in registers: the legs still move u, rhs and the
// transfer operand once, plus `a` once: 4.25 w / DoF per leg against 4 w per sweep with one launch per operator.
template <typename T, int HALO, bool PROLONG, int POST, bool ZERO_INIT, typename TX, typename TC, int TAG, int SM, int TI = kFusedTI,
          bool VAR = false>
__global__ __launch_bounds__(kFusedBlock) void fused_jacobi_kernel(
    const T* __restrict__ u, const T* __restrict__ rhs, T* __restrict__ out,
    const TX* __restrict__ e_coarse,      // PROLONG: coarse correction (dtype TX)
    TX* __restrict__ rhs_coarse,          // POST == kPostRestrict: coarse rhs (dtype TX)
    double* __restrict__ partials,        // POST == kPostNorm: one partial sum of r^2 (interior cells) per block
    FusedArgs a, T ihx2, T ihy2, T invD, T D, T omega, T one_m_omega, T coeff,
    const T* __restrict__ acoef,          // VAR: vertex values of the diffusion coefficient (same shape / pitch as u)
    T sigma,                              // VAR: Helmholtz shift added to the per-cell diagonal (constant path: folded into D)
    const T* __restrict__ rdiag) {        // VAR: reciprocal diagonal 1 / ((aip + aim) ihx2 + (ajp + ajm) ihy2 [+ sigma]) per cell (var_rdiag_kernel)
  using S = FusedShape<T, HALO, TI>;
  constexpr int N = S::N;
  __shared__ __attribute__((aligned(16))) T bufA[S::ELEMS];
  __shared__ __attribute__((aligned(16))) T bufB[S::ELEMS];
  __shared__ double red[kFusedBlock / 64];

  const int L = xcd_remap(blockIdx.x, a.ntiles);
  const int ti = L / a.tiles_j, tj = L - ti * a.tiles_j;
  const int i0 = 1 + ti * TI, j0 = tj * S::TJ;
  const int ri0 = i0 - HALO, rj0 = j0 - S::HV * N;           // global coords of region cell (0,0)
  if (a.select != 0) {
    const bool inner = ri0 >= a.in_i_lo && ri0 + S::RI <= a.in_i_hi && rj0 >= a.in_j_lo && rj0 + S::RJ <= a.in_j_hi;
    if ((a.select == 1) != inner) {
      if (POST == kPostNorm && threadIdx.x == 0) partials[blockIdx.x] = 0.0;
      return;
    }
  }
  const int cv = threadIdx.x % S::VPR, rg = threadIdx.x / S::VPR;
  const bool worker = rg < S::RG;
  const int gj0 = rj0 + cv * N;
  const int lc = cv * N;
  const int r_base = rg * S::RPT;

  // ---- VAR: face coefficients of this thread's cells into registers (vertex values staged through bufA) ----------
  constexpr int AVK = VAR ? S::RPT + 1 : 1, AVN = VAR ? N : 1, AHK = VAR ? S::RPT : 1, AHN = VAR ? N + 1 : 1;
  T av[AVK][AVN];     // av[k][e]: face between rows r_base + k - 1 and r_base + k     (a(i-1/2) of row k, a(i+1/2) of row k - 1)
  T ah[AHK][AHN];     // ah[k][e]: face between columns lc + e - 1 and lc + e of row r_base + k
  T rdv[AHK][AVN];    // rdv[k][e]: reciprocal diagonal of the thread's cells (the level's rdiag field)
  if (VAR) {
#pragma unroll
    for (int k = 0; k < S::RPT; ++k) {
      const int r = r_base + k, gi = ri0 + r;
      Pack<T> p = zero_pack<T>();
      if (worker && r < S::RI && gi >= 0 && gi < a.nx && gj0 >= 0 && gj0 < a.nyv) p = ldg(rdiag + (size_t)gi * a.ld + gj0);
#pragma unroll
      for (int e = 0; e < AVN; ++e) rdv[VAR ? k : 0][VAR ? e : 0] = p.v[VAR ? e : 0];
    }
#pragma unroll
    for (int k = 0; k < S::RPT; ++k) {
      const int r = r_base + k, gi = ri0 + r;
      if (!worker || r >= S::RI) continue;
      Pack<T> p = zero_pack<T>();
      if (gi >= 0 && gi < a.nx && gj0 >= 0 && gj0 < a.nyv) p = ldg(acoef + (size_t)gi * a.ld + gj0);
      *reinterpret_cast<Pack<T>*>(bufA + r * S::RJ + lc) = p;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < AVK; ++k)
#pragma unroll
      for (int e = 0; e < AVN; ++e) av[k][e] = T(0);
#pragma unroll
    for (int k = 0; k < AHK; ++k)
#pragma unroll
      for (int e = 0; e < AHN; ++e) ah[k][e] = T(0);
    if (worker && r_base < S::RI) {
      Pack<T> up = (r_base >= 1) ? *reinterpret_cast<const Pack<T>*>(bufA + (r_base - 1) * S::RJ + lc) : zero_pack<T>();
      Pack<T> mid = *reinterpret_cast<const Pack<T>*>(bufA + r_base * S::RJ + lc);
#pragma unroll
      for (int k = 0; k < S::RPT; ++k) {
        const int r = r_base + k;
        if (r >= S::RI) break;
        const Pack<T> dn = (r + 1 < S::RI) ? *reinterpret_cast<const Pack<T>*>(bufA + (r + 1) * S::RJ + lc) : zero_pack<T>();
        if (r >= 1 && r < S::RI - 1) {          // region-edge rows are never updated: their faces are not needed
          const T left = bufA[r * S::RJ + lc - 1];
          const T right = bufA[r * S::RJ + lc + N];
#pragma unroll
          for (int e = 0; e < N; ++e) {
            av[VAR ? k : 0][VAR ? e : 0] = T(0.5) * (mid.v[e] + up.v[e]);
            av[VAR ? k + 1 : 0][VAR ? e : 0] = T(0.5) * (mid.v[e] + dn.v[e]);
            ah[VAR ? k : 0][VAR ? e : 0] = T(0.5) * (mid.v[e] + ((e == 0) ? left : mid.v[e > 0 ? e - 1 : 0]));
          }
          ah[VAR ? k : 0][VAR ? N : 0] = T(0.5) * (mid.v[N - 1] + right);
        }
        up = mid;
        mid = dn;
      }
    }
    __syncthreads();
  }

  // ---- load: rhs strip into registers, u (+ P e) into LDS --------------------------------------
  // PROLONG: the (RI/2 + 2) x (RJ/2 + 2) patch of coarse values under the region is staged once, coalesced, in
  // bufB (free until the first sweep writes it); the interpolation then reads LDS instead of issuing 2 (N/2 + 1)
  // scalar global loads per row vector and thread -- three times the load instructions of u and rhs together.
  constexpr int PH = S::RI / 2 + 2, PW = S::RJ / 2 + 2;
  static_assert(!PROLONG || (size_t)PH * PW * sizeof(TX) <= sizeof(T) * S::ELEMS, "coarse patch does not fit the LDS buffer");
  const int pic0 = (ri0 >> 1) + a.ci_off, pjc0 = (rj0 >> 1) + a.cj_off;     // coarse cell of patch entry (0, 0)
  TX* patch = reinterpret_cast<TX*>(bufB);
  if (PROLONG && !MG_EXP_NO_LOAD) {
    for (int idx = threadIdx.x; idx < PH * PW; idx += kFusedBlock) {
      const int pr = idx / PW, pc = idx - pr * PW;
      const int ic = pic0 + pr, jc = pjc0 + pc;
      patch[idx] = (ic >= 0 && ic < a.nxc && jc >= 0 && jc < a.nyc) ? e_coarse[(size_t)ic * a.ldc + jc] : TX(0);
    }
  }
  Pack<T> f[S::RPT], uu[S::RPT];
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const int r = r_base + k, gi = ri0 + r;
    f[k] = zero_pack<T>();
    uu[k] = zero_pack<T>();
    if (!worker || r >= S::RI) continue;
    if (gi >= 0 && gi < a.nx && gj0 >= 0 && gj0 < a.nyv && !MG_EXP_NO_LOAD) {
      f[k] = ldg(rhs + (size_t)gi * a.ld + gj0);
      if (!ZERO_INIT) uu[k] = ldg(u + (size_t)gi * a.ld + gj0);
    }
  }
  if (PROLONG) __syncthreads();
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const int r = r_base + k, gi = ri0 + r;
    if (!worker || r >= S::RI) continue;
    if (PROLONG && gi >= 0 && gi < a.nx && gj0 >= 0 && gj0 < a.nyv && !MG_EXP_NO_LOAD) {
      using TS = typename std::conditional<(sizeof(TC) > sizeof(T)), TC, T>::type;
      TC val[N];
      bool ok[N];
      // the patch addressed like the coarse array: entry (ic, jc) at pe[ic * PW + jc]
      const TX* pe = patch - ((ptrdiff_t)pic0 * PW + pjc0);
      prolong_vec<TX, TC, N>(pe, PW, a.nxc, a.nyc, gi, gj0, a.nx, a.ny, a.sides, val, ok, a.ci_off, a.cj_off);
#pragma unroll
      for (int e = 0; e < N; ++e)
        if (ok[e]) uu[k].v[e] = (T)((TS)uu[k].v[e] + (TS)val[e]);
    }
    *reinterpret_cast<Pack<T>*>(bufA + r * S::RJ + lc) = uu[k];
  }
  __syncthreads();

  // ---- nsweep Jacobi sweeps, ping-pong between the two LDS buffers ------------------------------
  // A thread's RPT rows are consecutive: the row above / current / below slide through registers, so a sweep
  // reads RPT + 2 row vectors (not 3 RPT) plus the two lateral scalars per row.
  T* src = bufA;
  T* dst = bufB;
  if (SM == kSmRbgs) {
    // Red-black Gauss-Seidel IN PLACE in bufA: colour 0 = (i+j) even first (solvers/smoothers.py:183-205).  A colour
    // pass reads only the other colour (plus the cell itself) and rewrites other-colour cells with the values it
    // read, so concurrent vector stores never change what a neighbour reads.  Each pass costs one halo cell.
    for (int s = 0; s < 2 * a.nsweep; ++s) {
      const int colour = s & 1;
#pragma unroll
      for (int k = 0; k < S::RPT; ++k) {
        const int r = r_base + k, gi = ri0 + r;
        if (!worker || r >= S::RI) continue;
        if (r >= 1 && r < S::RI - 1 && gi >= 1 && gi < a.nx - 1) {
          const Pack<T> up = *reinterpret_cast<const Pack<T>*>(src + (r - 1) * S::RJ + lc);
          const Pack<T> mid = *reinterpret_cast<const Pack<T>*>(src + r * S::RJ + lc);
          const Pack<T> dn = *reinterpret_cast<const Pack<T>*>(src + (r + 1) * S::RJ + lc);
          const T left = src[r * S::RJ + lc - 1];
          const T right = src[r * S::RJ + lc + N];
          Pack<T> o = mid;
#pragma unroll
          for (int e = 0; e < N; ++e) {
            const T w = (e == 0) ? left : mid.v[e - 1];
            const T ea = (e == N - 1) ? right : mid.v[e + 1];
            T un;
            if (VAR) {
              const T aip = av[VAR ? k + 1 : 0][VAR ? e : 0], aim = av[VAR ? k : 0][VAR ? e : 0];
              const T ajp = ah[VAR ? k : 0][VAR ? e + 1 : 0], ajm = ah[VAR ? k : 0][VAR ? e : 0];
              const T sx = aip * dn.v[e] + aim * up.v[e], sy = ajp * ea + ajm * w;
              un = (f[k].v[e] + (ihx2 * sx + ihy2 * sy)) * rdv[VAR ? k : 0][VAR ? e : 0];
            } else {
              const T nb = ihx2 * (dn.v[e] + up.v[e]) + ihy2 * (ea + w);
              un = a.use_div ? (f[k].v[e] + nb) / D : (f[k].v[e] + nb) * invD;
            }
            const T res = one_m_omega * mid.v[e] + omega * un;
            const int gj = gj0 + e;
            if (gj >= 1 && gj < a.ny - 1 && (((gi + gj + a.colour_offset) & 1) == colour)) o.v[e] = res;
          }
          *reinterpret_cast<Pack<T>*>(src + r * S::RJ + lc) = o;
        }
      }
      __syncthreads();
    }
  } else
  for (int s = 0; s < (MG_EXP_NO_COMPUTE ? 0 : a.nsweep); ++s) {
    if (worker && r_base < S::RI) {
      Pack<T> up = (r_base >= 1) ? *reinterpret_cast<const Pack<T>*>(src + (r_base - 1) * S::RJ + lc) : zero_pack<T>();
      Pack<T> mid = *reinterpret_cast<const Pack<T>*>(src + r_base * S::RJ + lc);
#pragma unroll
      for (int k = 0; k < S::RPT; ++k) {
        const int r = r_base + k, gi = ri0 + r;
        if (r >= S::RI) break;
        const Pack<T> dn = (r + 1 < S::RI) ? *reinterpret_cast<const Pack<T>*>(src + (r + 1) * S::RJ + lc) : zero_pack<T>();
        Pack<T> o = mid;
        if (r >= 1 && r < S::RI - 1 && gi >= 1 && gi < a.nx - 1) {
          const T left = src[r * S::RJ + lc - 1];
          const T right = src[r * S::RJ + lc + N];
#pragma unroll
          for (int e = 0; e < N; ++e) {
            const T w = (e == 0) ? left : mid.v[e - 1];
            const T ea = (e == N - 1) ? right : mid.v[e + 1];
            T un;
            if (VAR) {
              const T aip = av[VAR ? k + 1 : 0][VAR ? e : 0], aim = av[VAR ? k : 0][VAR ? e : 0];
              const T ajp = ah[VAR ? k : 0][VAR ? e + 1 : 0], ajm = ah[VAR ? k : 0][VAR ? e : 0];
              const T sx = aip * dn.v[e] + aim * up.v[e], sy = ajp * ea + ajm * w;
              un = (f[k].v[e] + (ihx2 * sx + ihy2 * sy)) * rdv[VAR ? k : 0][VAR ? e : 0];
            } else {
              const T nb = ihx2 * (dn.v[e] + up.v[e]) + ihy2 * (ea + w);
              un = a.use_div ? (f[k].v[e] + nb) / D : (f[k].v[e] + nb) * invD;
            }
            const T res = one_m_omega * mid.v[e] + omega * un;
            const int gj = gj0 + e;
            if (gj >= 1 && gj < a.ny - 1) o.v[e] = res;
          }
        }
        *reinterpret_cast<Pack<T>*>(dst + r * S::RJ + lc) = o;
        up = mid;
        mid = dn;
      }
    }
    __syncthreads();
    T* t = src; src = dst; dst = t;
  }
  // `src` now holds the smoothed iterate (exact on the tile and on a ring of HALO - nsweep cells)

  // ---- write the tile of u' ------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < S::RPT; ++k) {
    const int r = r_base + k, gi = ri0 + r;
    if (!worker || r >= S::RI) continue;
    if (!MG_EXP_NO_STORE && r >= HALO && r < HALO + TI && cv >= S::HV && cv < S::HV + S::TJ / N && gi < a.nx && gj0 < a.nyv)
      stg(out + (size_t)gi * a.ld + gj0, *reinterpret_cast<const Pack<T>*>(src + r * S::RJ + lc));
  }

  if (POST == kPostNone) return;

  // ---- residual on the region (r = f on boundary cells, 0 outside the grid) -------------------------
  double acc = 0.0;
  if (worker && r_base < S::RI) {
    Pack<T> up = (r_base >= 1) ? *reinterpret_cast<const Pack<T>*>(src + (r_base - 1) * S::RJ + lc) : zero_pack<T>();
    Pack<T> mid = *reinterpret_cast<const Pack<T>*>(src + r_base * S::RJ + lc);
#pragma unroll
    for (int k = 0; k < S::RPT; ++k) {
      const int r = r_base + k, gi = ri0 + r;
      if (r >= S::RI) break;
      const Pack<T> dn = (r + 1 < S::RI) ? *reinterpret_cast<const Pack<T>*>(src + (r + 1) * S::RJ + lc) : zero_pack<T>();
      Pack<T> o = f[k];
      const bool in_tile = r >= HALO && r < HALO + TI && cv >= S::HV && cv < S::HV + S::TJ / N;
      const bool wanted = (POST == kPostRestrict) || in_tile;      // the norm only needs r on the tile itself
      if (!MG_EXP_NO_COMPUTE && wanted && r >= 1 && r < S::RI - 1 && gi >= 1 && gi < a.nx - 1) {
        const T left = src[r * S::RJ + lc - 1];
        const T right = src[r * S::RJ + lc + N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const T w = (e == 0) ? left : mid.v[e - 1];
          const T ea = (e == N - 1) ? right : mid.v[e + 1];
          T au;
          if (VAR) {
            const T aip = av[VAR ? k + 1 : 0][VAR ? e : 0], aim = av[VAR ? k : 0][VAR ? e : 0];
            const T ajp = ah[VAR ? k : 0][VAR ? e + 1 : 0], ajm = ah[VAR ? k : 0][VAR ? e : 0];
            const T sx = aip * dn.v[e] + aim * up.v[e], sy = ajp * ea + ajm * w;
            const T D0 = (aip + aim) * ihx2 + (ajp + ajm) * ihy2;
            au = coeff * ((sx * ihx2 + sy * ihy2) - mid.v[e] * ((sigma != T(0)) ? D0 + sigma : D0));
          } else {
            au = coeff * (((dn.v[e] + up.v[e]) * ihx2 + (ea + w) * ihy2) - mid.v[e] * D);
          }
          const int gj = gj0 + e;
          if (gj >= 1 && gj < a.ny - 1) {
            o.v[e] = f[k].v[e] - au;
            if (POST == kPostNorm && gi >= a.ni_lo && gi < a.ni_hi && gj >= a.nj_lo && gj < a.nj_hi) acc += (double)o.v[e] * (double)o.v[e];
          }
        }
      }
      if (POST == kPostRestrict) *reinterpret_cast<Pack<T>*>(dst + r * S::RJ + lc) = o;
      up = mid;
      mid = dn;
    }
  }

  if (POST == kPostNorm) {
    const double t = block_reduce_sum<kFusedBlock / 64>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
    return;
  }

  // ---- full-weighting restriction of the interior coarse cells that sit on this tile ---------------
  __syncthreads();
  constexpr int CI = TI / 2, CJ = S::TJ / 2;                // coarse cells per tile
  for (int c = threadIdx.x; c < CI * CJ; c += kFusedBlock) {
    const int ci = c / CJ, cj = c - ci * CJ;
    const int fi = i0 + 1 + 2 * ci, fj = j0 + 2 * cj;  // i0 is odd: even fine rows are i0+1, i0+3, ...
    const int ic = (fi >> 1) + a.ci_off, jc = (fj >> 1) + a.cj_off;
    if (ic < 1 || ic > a.nxc - 2 || jc < 1 || jc > a.nyc - 2) continue;
    if (fi < 1 || fi > a.nx - 2 || fj < 1 || fj > a.ny - 2) continue;     // coarse cell without a full fine neighbourhood here
    const T* p = dst + (fi - ri0) * S::RJ + (fj - rj0);
    const T corners = ((p[-S::RJ - 1] + p[-S::RJ + 1]) + p[S::RJ - 1]) + p[S::RJ + 1];
    const T edges = ((p[-S::RJ] + p[S::RJ]) + p[-1]) + p[1];
    if (MG_EXP_NO_STORE) { if (corners == T(12345.678)) rhs_coarse[0] = (TX)edges; continue; }
    rhs_coarse[(size_t)ic * a.ldc + jc] = (TX)((T(1.0 / 16.0) * corners + T(1.0 / 8.0) * edges) + T(1.0 / 4.0) * p[0]);
  }
}

// Injection of the boundary ring fine -> coarse (the boundary part of operators/transfer.py:109-113).  With
// r = f on boundary cells this ring of every coarse rhs is constant over a solve: written once per rhs.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void inject_ring_kernel(const TIN* __restrict__ fine, TOUT* __restrict__ coarse,
                                                             int nxf, int nyf, int ldf, int nxc, int nyc, int ldc, int sides,
                                                             int ci_off, int cj_off) {
  const int ring = 2 * nxc + 2 * nyc;
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < ring; t += gridDim.x * kBlock) {
    int ic, jc, side;
    if (t < nyc) { ic = 0; jc = t; side = kSideILo; }
    else if (t < 2 * nyc) { ic = nxc - 1; jc = t - nyc; side = kSideIHi; }
    else if (t < 2 * nyc + nxc) { ic = t - 2 * nyc; jc = 0; side = kSideJLo; }
    else { ic = t - 2 * nyc - nxc; jc = nyc - 1; side = kSideJHi; }
    if (!(sides & side)) continue;                       // a ghost edge of a sub-domain: not a boundary
    const int fi = 2 * (ic - ci_off), fj = 2 * (jc - cj_off);
    if (fi < 0 || fi >= nxf || fj < 0 || fj >= nyf) continue;
    coarse[(size_t)ic * ldc + jc] = (TOUT)fine[(size_t)fi * ldf + fj];
  }
}

}  // namespace mg
