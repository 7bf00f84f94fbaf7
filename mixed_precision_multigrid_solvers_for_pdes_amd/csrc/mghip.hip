// libmghip.so -- C ABI (include/mghip.h) over the CDNA4 kernels in mg_kernels.hpp, plus the
// device-resident V/W/F-cycle driver (reference: solvers/multigrid.py:184-337, gpu/gpu_solver.py:186-446).
// No Python, no torch types: plain pointers and sizes.
#include "mg_host.hpp"
#include "mg_kernels.hpp"
#include "mg_rb_kernels.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mghip.h"

using namespace mgh;

namespace {

thread_local std::string g_last_error;

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int fail(std::string* where, int code, const std::string& msg) {
  g_last_error = msg;
  if (where) *where = msg;
  return code;
}

#define HIPC(errstr, call)                                                                         \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(errstr, (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? MG_ERR_NO_DEVICE : \
                          (e_ == hipErrorOutOfMemory ? MG_ERR_ALLOC : MG_ERR_HIP),                 \
                  std::string(#call) + ": " + hipGetErrorString(e_));                              \
  } while (0)

// Row pitch in elements: rows start on 512-byte boundaries (every tile row segment is line aligned).
inline int pitch_elems(int dt, int ny) {
  const size_t bytes = ((size_t)ny * esize(dt) + 511) / 512 * 512;
  return (int)(bytes / esize(dt));
}


template <typename T>
mg::TileGeom make_geom(int nx, int ny, int ld, bool interior_only) {
  using S = mg::TileShape<T>;
  mg::TileGeom g;
  g.nx = nx; g.ny = ny; g.ld = ld;
  g.nyv = std::min(ld, (ny + S::N - 1) / S::N * S::N);
  g.i_org = interior_only ? 1 : 0;
  const int rows = interior_only ? nx - 2 : nx;
  const int cols = interior_only ? ny - 1 : ny;     // column ny-1 is boundary: never written by a smoother
  const int tiles_i = (rows + mg::kTI - 1) / mg::kTI;
  g.tiles_j = (cols + S::TJ - 1) / S::TJ;
  g.ntiles = tiles_i * g.tiles_j;
  return g;
}

// Upper bound of the per-workgroup partial sums any norm launch on an (nx, ny) field writes: the grid-stride
// reductions use <= 2048 workgroups, the residual+norm kernel one per kTI-row tile, the fused up leg one per tile of
// its own (shorter) tile height; fp64 tiles are the narrowest (64 columns).
inline size_t max_partials(int nx, int ny) {
  const long long tj = (ny + 63) / 64 + 1;
  const int ti_min = std::min(mg::kTI, std::min(mg::kFusedTI, std::min(mg::kFusedTISmall, mg::kFusedTITiny)));
  const long long ti = (nx + ti_min - 1) / ti_min + 1;
  return (size_t)std::max<long long>(2048, ti * tj);
}

inline int grid_for(long long work_items) {
  long long b = (work_items + mg::kBlock - 1) / mg::kBlock;
  return (int)std::max<long long>(1, std::min<long long>(b, 256 * 16));
}

// ------------------------------------------------------------------ typed launchers ------------
template <typename T>
void launch_jacobi(const void* u, const void* rhs, void* out, int nx, int ny, int ld, double hx, double hy,
                   double omega, hipStream_t st, bool fine = false, double sigma = 0.0) {
  if (nx < 3 || ny < 3) return;
  const Coef c = coefs(hx, hy, sigma);
  const mg::TileGeom g = make_geom<T>(nx, ny, ld, true);
  auto k = c.pow2 ? (fine ? mg::jacobi_kernel<T, mg::kFineTag, false> : mg::jacobi_kernel<T, mg::kCoarseTag, false>)
                  : (fine ? mg::jacobi_kernel<T, mg::kFineTag, true> : mg::jacobi_kernel<T, mg::kCoarseTag, true>);
  hipLaunchKernelGGL(k, dim3(g.ntiles), dim3(mg::kBlock), 0, st, (const T*)u, (const T*)rhs, (T*)out,
                     g, (T)c.ihx2, (T)c.ihy2, (T)c.invD, (T)c.diag, (T)omega, (T)(1.0 - omega));
}

template <typename T>
void launch_rbgs_colour(void* u, const void* rhs, int nx, int ny, int ld, double hx, double hy, double omega,
                        int colour, int poff, hipStream_t st, bool fine = false, double sigma = 0.0) {
  if (nx < 3 || ny < 3) return;
  const Coef c = coefs(hx, hy, sigma);
  const mg::TileGeom g = make_geom<T>(nx, ny, ld, true);
  auto k = c.pow2 ? (fine ? mg::rbgs_colour_kernel<T, mg::kFineTag, false> : mg::rbgs_colour_kernel<T, mg::kCoarseTag, false>)
                  : (fine ? mg::rbgs_colour_kernel<T, mg::kFineTag, true> : mg::rbgs_colour_kernel<T, mg::kCoarseTag, true>);
  hipLaunchKernelGGL(k, dim3(g.ntiles), dim3(mg::kBlock), 0, st, (T*)u, (const T*)rhs, g,
                     (T)c.ihx2, (T)c.ihy2, (T)c.invD, (T)c.diag, (T)omega, (T)(1.0 - omega), colour, poff & 1);
}

// returns the number of partials written (0 when NORM is off)
template <typename T, bool WRITE_R, bool NORM>
int launch_residual(const void* u, const void* f, void* r, double* partials, int nx, int ny, int ld, double hx,
                    double hy, double coeff, hipStream_t st, bool fine = false, double sigma = 0.0) {
  const Coef c = coefs(hx, hy, sigma);
  const mg::TileGeom g = make_geom<T>(nx, ny, ld, false);
  auto k = fine ? mg::residual_kernel<T, WRITE_R, NORM, mg::kFineTag> : mg::residual_kernel<T, WRITE_R, NORM, mg::kCoarseTag>;
  hipLaunchKernelGGL(k, dim3(g.ntiles), dim3(mg::kBlock), 0, st, (const T*)u,
                     (const T*)f, (T*)r, partials, g, (T)c.ihx2, (T)c.ihy2, (T)c.diag, (T)coeff);
  return NORM ? g.ntiles : 0;
}

template <typename T>
int launch_sumsq(const void* x, double* partials, int ld, int i_lo, int i_hi, int j_lo, int j_hi, hipStream_t st) {
  const int N = mg::VecW<T>::N;
  const long long vecs = (long long)std::max(0, i_hi - i_lo) * ((j_hi + N - 1) / N - j_lo / N);
  const int nb = std::min(grid_for(vecs), 2048);
  hipLaunchKernelGGL(mg::sumsq_kernel<T>, dim3(nb), dim3(mg::kBlock), 0, st, (const T*)x, partials, ld, i_lo, i_hi, j_lo,
                     j_hi);
  return nb;
}

inline void launch_reduce(const double* partials, int n, double* out, hipStream_t st, mg::HostMailbox* mailbox = nullptr,
                          unsigned long long seq = 0) {
  hipLaunchKernelGGL(mg::reduce_partials_kernel<0>, dim3(1), dim3(mg::kReduceBlock), 0, st, partials, n, out, mailbox, seq);
}

template <typename TI, typename TO>
void launch_restrict(const void* fine, void* coarse, int ldf, int nxc, int nyc, int ldc, int sides, hipStream_t st) {
  const int NO = mg::VecW<TO>::N;
  hipLaunchKernelGGL((mg::restrict_fw_kernel<TI, TO>), dim3(grid_for((long long)nxc * ((nyc + NO - 1) / NO))),
                     dim3(mg::kBlock), 0, st, (const TI*)fine, (TO*)coarse, ldf, nxc, nyc, ldc, sides);
}

template <typename TCI, typename TF, typename TC, bool ADD>
void launch_prolong(const void* e, void* u, int nxf, int nyf, int ldf, int nxc, int nyc, int ldc, int sides,
                    hipStream_t st) {
  const int N = mg::VecW<TF>::N;
  hipLaunchKernelGGL((mg::prolong_kernel<TCI, TF, TC, ADD>), dim3(grid_for((long long)nxf * ((nyf + N - 1) / N))),
                     dim3(mg::kBlock), 0, st, (const TCI*)e, (TF*)u, nxf, nyf, ldf, nxc, nyc, ldc, sides);
}

template <typename TI, typename TO>
void launch_convert(const void* in, void* out, int nx, int ny, int ldi, int ldo, hipStream_t st) {
  hipLaunchKernelGGL((mg::convert_kernel<TI, TO>), dim3(grid_for((long long)nx * ny)), dim3(mg::kBlock), 0, st,
                     (const TI*)in, (TO*)out, nx, ny, ldi, ldo);
}

template <typename T>
void launch_coarse(void* u, const void* rhs, int nx, int ny, int ld, double hx, double hy, double coeff, double omega,
                   double tol, int maxit, int* sweeps_dev, hipStream_t st, bool zero_init = false, const void* a = nullptr,
                   double sigma = 0.0) {
  const Coef c = coefs(hx, hy, sigma);
  if (a) {          // variable coefficient: the general one-workgroup kernel
    if (zero_init) (void)hipMemsetAsync(u, 0, (size_t)nx * ld * sizeof(T), st);
    hipLaunchKernelGGL(mg::coarse_lexgs_kernel<T>, dim3(1), dim3(mg::kBlock), 0, st, (T*)u, (const T*)rhs, nx, ny, ld,
                       (T)(hx * hx), (T)(hy * hy), (T)omega, (T)(1.0 - omega), (T)c.diag, (T)coeff, hx * hy, tol, maxit,
                       sweeps_dev, (const T*)a, (T)sigma);
    return;
  }
  if (nx * ny <= mg::kCoarseLdsCells) {
    hipLaunchKernelGGL(mg::coarse_lexgs_small_kernel<T>, dim3(1), dim3(64), 0, st, (T*)u, (const T*)rhs, nx, ny, ld,
                       (T)(hx * hx), (T)(hy * hy), (T)omega, (T)(1.0 - omega), (T)c.diag, (T)coeff, hx * hy, tol, maxit,
                       sweeps_dev, zero_init ? 1 : 0, c.all_pow2 ? 1 : 0, sqrt_threshold(tol));
    return;
  }
  if (zero_init) (void)hipMemsetAsync(u, 0, (size_t)nx * ld * sizeof(T), st);
  hipLaunchKernelGGL(mg::coarse_lexgs_kernel<T>, dim3(1), dim3(mg::kBlock), 0, st, (T*)u, (const T*)rhs, nx, ny, ld,
                     (T)(hx * hx), (T)(hy * hy), (T)omega, (T)(1.0 - omega), (T)c.diag, (T)coeff, hx * hy, tol,
                     maxit, sweeps_dev, (const T*)nullptr);
}

// ------------------------------------------------------------------ dtype dispatch --------------
bool jacobi_rb(int dt, const void* u, const void* rhs, void* out, int nx, int ny, int ld, double hx, double hy, double omega,
               hipStream_t st, double sigma);       // the register-blocked single sweep (defined with the other launchers below)
void d_jacobi(int dt, const void* u, const void* rhs, void* out, int nx, int ny, int ld, double hx, double hy,
              double omega, hipStream_t st, bool fine = false, double sigma = 0.0) {
  if (jacobi_rb(dt, u, rhs, out, nx, ny, ld, hx, hy, omega, st, sigma)) return;
  if (dt == MG_F32) launch_jacobi<float>(u, rhs, out, nx, ny, ld, hx, hy, omega, st, fine, sigma);
  else launch_jacobi<double>(u, rhs, out, nx, ny, ld, hx, hy, omega, st, fine, sigma);
}
void d_rbgs_colour(int dt, void* u, const void* rhs, int nx, int ny, int ld, double hx, double hy, double omega,
                   int colour, int poff, hipStream_t st, bool fine = false, double sigma = 0.0) {
  if (dt == MG_F32) launch_rbgs_colour<float>(u, rhs, nx, ny, ld, hx, hy, omega, colour, poff, st, fine, sigma);
  else launch_rbgs_colour<double>(u, rhs, nx, ny, ld, hx, hy, omega, colour, poff, st, fine, sigma);
}
void d_residual(int dt, const void* u, const void* f, void* r, int nx, int ny, int ld, double hx, double hy,
                double coeff, hipStream_t st, bool fine = false, double sigma = 0.0) {
  if (dt == MG_F32) launch_residual<float, true, false>(u, f, r, nullptr, nx, ny, ld, hx, hy, coeff, st, fine, sigma);
  else launch_residual<double, true, false>(u, f, r, nullptr, nx, ny, ld, hx, hy, coeff, st, fine, sigma);
}
int d_residual_norm(int dt, const void* u, const void* f, double* partials, int nx, int ny, int ld, double hx,
                    double hy, double coeff, hipStream_t st, bool fine = false, double sigma = 0.0) {
  if (dt == MG_F32) return launch_residual<float, false, true>(u, f, nullptr, partials, nx, ny, ld, hx, hy, coeff, st, fine, sigma);
  return launch_residual<double, false, true>(u, f, nullptr, partials, nx, ny, ld, hx, hy, coeff, st, fine, sigma);
}
int d_sumsq(int dt, const void* x, double* partials, int ld, int i_lo, int i_hi, int j_lo, int j_hi, hipStream_t st) {
  return dt == MG_F32 ? launch_sumsq<float>(x, partials, ld, i_lo, i_hi, j_lo, j_hi, st)
                      : launch_sumsq<double>(x, partials, ld, i_lo, i_hi, j_lo, j_hi, st);
}
// whole-grid form: coarse dims follow from the fine ones and all four edges are physical boundaries
void d_restrict_sub(int di, int dout, const void* fine, void* coarse, int ldf, int nxc, int nyc, int ldc, int sides,
                    hipStream_t st) {
  if (di == MG_F32 && dout == MG_F32) launch_restrict<float, float>(fine, coarse, ldf, nxc, nyc, ldc, sides, st);
  else if (di == MG_F64 && dout == MG_F64) launch_restrict<double, double>(fine, coarse, ldf, nxc, nyc, ldc, sides, st);
  else if (di == MG_F64 && dout == MG_F32) launch_restrict<double, float>(fine, coarse, ldf, nxc, nyc, ldc, sides, st);
  else launch_restrict<float, double>(fine, coarse, ldf, nxc, nyc, ldc, sides, st);
}
void d_restrict(int di, int dout, const void* fine, void* coarse, int nxf, int nyf, int ldf, int ldc, hipStream_t st) {
  d_restrict_sub(di, dout, fine, coarse, ldf, (nxf - 1) / 2 + 1, (nyf - 1) / 2 + 1, ldc, mg::kAllSides, st);
}
// dc: dtype of the coarse field, df: of the fine field, dcomp: interpolation arithmetic (the fine GRID's dtype)
template <bool ADD>
int d_prolong_sub(int dc, int df, int dcomp, const void* e, void* u, int nxf, int nyf, int ldf, int nxc, int nyc, int ldc,
                  int sides, hipStream_t st) {
  if (dcomp == MG_F32) {
    if (dc == MG_F32 && df == MG_F32) { launch_prolong<float, float, float, ADD>(e, u, nxf, nyf, ldf, nxc, nyc, ldc, sides, st); return MG_OK; }
    return MG_ERR_INVALID_VALUE;   // fp32 interpolation only exists for an all-fp32 grid
  }
  if (dc == MG_F64 && df == MG_F64) launch_prolong<double, double, double, ADD>(e, u, nxf, nyf, ldf, nxc, nyc, ldc, sides, st);
  else if (dc == MG_F32 && df == MG_F64) launch_prolong<float, double, double, ADD>(e, u, nxf, nyf, ldf, nxc, nyc, ldc, sides, st);
  else if (dc == MG_F64 && df == MG_F32) launch_prolong<double, float, double, ADD>(e, u, nxf, nyf, ldf, nxc, nyc, ldc, sides, st);
  else launch_prolong<float, float, double, ADD>(e, u, nxf, nyf, ldf, nxc, nyc, ldc, sides, st);
  return MG_OK;
}
template <bool ADD>
int d_prolong(int dc, int df, int dcomp, const void* e, void* u, int nxf, int nyf, int ldf, int ldc, hipStream_t st) {
  return d_prolong_sub<ADD>(dc, df, dcomp, e, u, nxf, nyf, ldf, (nxf - 1) / 2 + 1, (nyf - 1) / 2 + 1, ldc, mg::kAllSides, st);
}
void d_convert(int di, int dout, const void* in, void* out, int nx, int ny, int ldi, int ldo, hipStream_t st) {
  if (di == MG_F32 && dout == MG_F32) launch_convert<float, float>(in, out, nx, ny, ldi, ldo, st);
  else if (di == MG_F64 && dout == MG_F64) launch_convert<double, double>(in, out, nx, ny, ldi, ldo, st);
  else if (di == MG_F64 && dout == MG_F32) launch_convert<double, float>(in, out, nx, ny, ldi, ldo, st);
  else launch_convert<float, double>(in, out, nx, ny, ldi, ldo, st);
}
template <typename TI, typename TO>
void launch_convert_ring(const void* in, void* out, int nx, int ny, int ldi, int ldo, hipStream_t st) {
  hipLaunchKernelGGL((mg::convert_ring_kernel<TI, TO>), dim3(grid_for(2LL * nx + 2LL * ny)), dim3(mg::kBlock), 0, st,
                     (const TI*)in, (TO*)out, nx, ny, ldi, ldo);
}
// boundary ring of `in` -> boundary ring of `out` (the interior of `out` is left alone)
void d_convert_ring(int di, int dout, const void* in, void* out, int nx, int ny, int ldi, int ldo, hipStream_t st) {
  if (di == MG_F32 && dout == MG_F32) launch_convert_ring<float, float>(in, out, nx, ny, ldi, ldo, st);
  else if (di == MG_F64 && dout == MG_F64) launch_convert_ring<double, double>(in, out, nx, ny, ldi, ldo, st);
  else if (di == MG_F64 && dout == MG_F32) launch_convert_ring<double, float>(in, out, nx, ny, ldi, ldo, st);
  else launch_convert_ring<float, double>(in, out, nx, ny, ldi, ldo, st);
}
void d_zero_interior(int dt, void* u, int nx, int ny, int ld, hipStream_t st) {
  if (nx < 3 || ny < 3) return;
  const long long vecs = (long long)(nx - 2) * ((ny + (int)(16 / esize(dt)) - 1) / (int)(16 / esize(dt)));
  if (dt == MG_F32) hipLaunchKernelGGL(mg::zero_interior_kernel<float>, dim3(grid_for(vecs)), dim3(mg::kBlock), 0, st, (float*)u, nx, ny, ld);
  else hipLaunchKernelGGL(mg::zero_interior_kernel<double>, dim3(grid_for(vecs)), dim3(mg::kBlock), 0, st, (double*)u, nx, ny, ld);
}
void d_coarse(int dt, void* u, const void* rhs, int nx, int ny, int ld, double hx, double hy, double coeff,
              double omega, double tol, int maxit, int* sweeps_dev, hipStream_t st, bool zero_init = false,
              const void* a = nullptr, double sigma = 0.0) {
  if (dt == MG_F32) launch_coarse<float>(u, rhs, nx, ny, ld, hx, hy, coeff, omega, tol, maxit, sweeps_dev, st, zero_init, a, sigma);
  else launch_coarse<double>(u, rhs, nx, ny, ld, hx, hy, coeff, omega, tol, maxit, sweeps_dev, st, zero_init, a, sigma);
}



// ------------------------------------------------------------------ variable coefficient ------
template <typename T, int MODE>
int launch_var(const void* u, const void* a, const void* f, void* out, double* partials, int nx, int ny, int ld, double hx,
               double hy, double omega, double coeff, int colour, int poff, hipStream_t st, double sigma = 0.0) {
  if (nx < 3 || ny < 3) return 0;
  const Coef c = coefs(hx, hy);
  const bool interior_only = (MODE == mg::kVarJacobi || MODE == mg::kVarRbgs);
  const mg::TileGeom g = make_geom<T>(nx, ny, ld, interior_only);
  hipLaunchKernelGGL((mg::varcoef_kernel<T, MODE>), dim3(g.ntiles), dim3(mg::kBlock), 0, st, (const T*)u, (const T*)a,
                     (const T*)f, (T*)out, partials, g, (T)c.ihx2, (T)c.ihy2, (T)omega, (T)(1.0 - omega), (T)coeff, colour,
                     poff & 1, (T)sigma);
  return g.ntiles;
}
template <int MODE>
int d_var(int dt, const void* u, const void* a, const void* f, void* out, double* partials, int nx, int ny, int ld, double hx,
          double hy, double omega, double coeff, int colour, int poff, hipStream_t st, double sigma = 0.0) {
  return dt == MG_F32 ? launch_var<float, MODE>(u, a, f, out, partials, nx, ny, ld, hx, hy, omega, coeff, colour, poff, st, sigma)
                      : launch_var<double, MODE>(u, a, f, out, partials, nx, ny, ld, hx, hy, omega, coeff, colour, poff, st, sigma);
}
template <typename TI, typename TO>
void launch_inject(const void* fine, void* coarse, int ldf, int nxc, int nyc, int ldc, int stride, hipStream_t st) {
  hipLaunchKernelGGL((mg::inject_kernel<TI, TO>), dim3(grid_for((long long)nxc * nyc)), dim3(mg::kBlock), 0, st,
                     (const TI*)fine, (TO*)coarse, ldf, nxc, nyc, ldc, stride);
}
void d_inject(int di, int dout, const void* fine, void* coarse, int ldf, int nxc, int nyc, int ldc, int stride, hipStream_t st) {
  if (di == MG_F32 && dout == MG_F32) launch_inject<float, float>(fine, coarse, ldf, nxc, nyc, ldc, stride, st);
  else if (di == MG_F64 && dout == MG_F64) launch_inject<double, double>(fine, coarse, ldf, nxc, nyc, ldc, stride, st);
  else if (di == MG_F64 && dout == MG_F32) launch_inject<double, float>(fine, coarse, ldf, nxc, nyc, ldc, stride, st);
  else launch_inject<float, double>(fine, coarse, ldf, nxc, nyc, ldc, stride, st);
}

template <typename T>
void launch_rdiag(const void* a, void* rd, int nx, int ny, int ld, double hx, double hy, double sigma, hipStream_t st) {
  const Coef c = coefs(hx, hy);
  hipLaunchKernelGGL(mg::var_rdiag_kernel<T>, dim3(grid_for((long long)nx * ny)), dim3(mg::kBlock), 0, st, (const T*)a, (T*)rd, nx, ny, ld,
                     (T)c.ihx2, (T)c.ihy2, (T)sigma);
}
void d_rdiag(int dt, const void* a, void* rd, int nx, int ny, int ld, double hx, double hy, double sigma, hipStream_t st) {
  if (dt == MG_F32) launch_rdiag<float>(a, rd, nx, ny, ld, hx, hy, sigma, st);
  else launch_rdiag<double>(a, rd, nx, ny, ld, hx, hy, sigma, st);
}

// ------------------------------------------------------------------ fused legs ----------------
struct LegGeom;
template <typename T, int HALO, int TI>
mg::FusedArgs fused_args(int nx, int ny, int ld, int nsweep, bool use_div, int nxc, int nyc, int ldc, int poff) {
  using S = mg::FusedShape<T, HALO, TI>;
  mg::FusedArgs a;
  a.nx = nx; a.ny = ny; a.ld = ld;
  a.nyv = std::min(ld, (ny + S::N - 1) / S::N * S::N);
  const int tiles_i = (nx - 2 + TI - 1) / TI;
  a.tiles_j = (ny - 1 + S::TJ - 1) / S::TJ;
  a.ntiles = tiles_i * a.tiles_j;
  a.nsweep = nsweep; a.nsweep2 = 0; a.band = 1; a.use_div = use_div ? 1 : 0; a.colour_offset = poff & 1;
  a.nxc = nxc; a.nyc = nyc; a.ldc = ldc;
  a.ci_off = a.cj_off = 0; a.sides = mg::kAllSides;
  a.ni_lo = 1; a.ni_hi = nx - 1; a.nj_lo = 1; a.nj_hi = ny - 1;
  a.select = 0; a.in_i_lo = a.in_j_lo = 0; a.in_i_hi = a.in_j_hi = 0;
  static const int flags = exp_env("MG_EXP_FLAGS", 0);          // measurement builds only (mg_host.hpp: exp_env)
  a.exp_flags = flags;
  return a;
}

struct LegGeom {      // what every fused launch needs
  int nx, ny, ld, nxc, nyc, ldc;
  double hx, hy, omega, coeff;
  int nsweep, poff;
  bool fine;
  // sub-domain extras (defaults = whole grid)
  int ci_off = 0, cj_off = 0, sides = mg::kAllSides;
  int ni_lo = -1, ni_hi = -1, nj_lo = -1, nj_hi = -1;     // norm window; -1: the interior
  int select = 0, in_i_lo = 0, in_i_hi = 0, in_j_lo = 0, in_j_hi = 0;   // tile selection (see mg::FusedArgs)
  double sigma = 0.0;                                                    // Helmholtz shift (see coefs)
  const void* acoef = nullptr;                                           // variable coefficient: vertex values (dtype / pitch of u)
  const void* rdiag = nullptr;                                           // ... and its reciprocal diagonal per cell (var_rdiag_kernel)
  int rb = 0;                                                            // 1: register-blocked legs on the bandwidth-bound levels
};
inline void apply_sub(mg::FusedArgs& a, const LegGeom& g) {
  a.ci_off = g.ci_off; a.cj_off = g.cj_off; a.sides = g.sides;
  if (g.ni_lo >= 0) { a.ni_lo = g.ni_lo; a.ni_hi = g.ni_hi; a.nj_lo = g.nj_lo; a.nj_hi = g.nj_hi; }
  a.select = g.select; a.in_i_lo = g.in_i_lo; a.in_i_hi = g.in_i_hi; a.in_j_lo = g.in_j_lo; a.in_j_hi = g.in_j_hi;
}

// down leg: nsweep sweeps + residual + full-weighting restriction (interior coarse cells).  TX = coarse rhs dtype.
// Tile height by level size: 32 rows where the launch is bandwidth-bound, 16 where it is latency-bound (<= ~1025^2).
// Variable-coefficient red-black GS keeps 6 halo cells and the face means of every owned cell in registers: with 32-row
// tiles that is 176 VGPRs (one workgroup per CU); 16-row tiles stay at 100 (two).
inline bool small_tiles(const LegGeom& g, int sm = mg::kSmJacobi) {
  return (long long)g.nx * g.ny <= 1100LL * 1100LL || (g.acoef && sm == mg::kSmRbgs);
}
// 8-row tiles: legs on levels of <= ~520^2 cells (513^2 and below) (MG_EXP_TINY=n: up to n^2 cells, 0 keeps the 16-row tiles: experiments)
inline bool tiny_tiles(const LegGeom& g) {
  static const long long lim = exp_env("MG_EXP_TINY", 520);   // 0: off
  return lim > 0 && (long long)g.nx * g.ny <= lim * lim;
}

template <typename T, typename TX, int SM, int TI>
void launch_down_ti(const void* u, const void* rhs, void* out, void* rhs_c, const LegGeom& g, bool zero_init, hipStream_t st) {
  constexpr int HALO = 2 * mg::sweep_halo(SM) + 2;
  const Coef c = coefs(g.hx, g.hy, g.sigma);
  mg::FusedArgs a = fused_args<T, HALO, TI>(g.nx, g.ny, g.ld, g.nsweep, !c.pow2, g.nxc, g.nyc, g.ldc, g.poff);
  apply_sub(a, g);
  void (*k)(const T*, const T*, T*, const TX*, TX*, double*, mg::FusedArgs, T, T, T, T, T, T, T, const T*, T, const T*);
  if (g.acoef) {      // variable coefficient (one symbol for all levels: TAG 0)
    k = zero_init ? mg::fused_jacobi_kernel<T, HALO, false, mg::kPostRestrict, true, TX, T, 0, SM, TI, true>
                  : mg::fused_jacobi_kernel<T, HALO, false, mg::kPostRestrict, false, TX, T, 0, SM, TI, true>;
  } else if (zero_init) k = g.fine ? mg::fused_jacobi_kernel<T, HALO, false, mg::kPostRestrict, true, TX, T, 1, SM, TI>
                                   : mg::fused_jacobi_kernel<T, HALO, false, mg::kPostRestrict, true, TX, T, 0, SM, TI>;
  else k = g.fine ? mg::fused_jacobi_kernel<T, HALO, false, mg::kPostRestrict, false, TX, T, 1, SM, TI>
                  : mg::fused_jacobi_kernel<T, HALO, false, mg::kPostRestrict, false, TX, T, 0, SM, TI>;
  hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(mg::kFusedBlock), 0, st, (const T*)u, (const T*)rhs, (T*)out, (const TX*)nullptr,
                     (TX*)rhs_c, (double*)nullptr, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD, (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)g.coeff,
                     (const T*)g.acoef, (T)g.sigma, (const T*)g.rdiag);
}
template <typename T, typename TX, int SM>
void launch_down(const void* u, const void* rhs, void* out, void* rhs_c, const LegGeom& g, bool zero_init, hipStream_t st) {
  if (tiny_tiles(g)) launch_down_ti<T, TX, SM, mg::kFusedTITiny>(u, rhs, out, rhs_c, g, zero_init, st);
  else if (small_tiles(g, SM)) launch_down_ti<T, TX, SM, mg::kFusedTISmall>(u, rhs, out, rhs_c, g, zero_init, st);
  else launch_down_ti<T, TX, SM, mg::kFusedTI>(u, rhs, out, rhs_c, g, zero_init, st);
}

// up leg: u += P e, nsweep sweeps, optional sum of r^2 over interior cells.  TX = coarse e dtype, TC = interpolation dtype.
// returns the number of partials (0 without norm)
template <typename T, typename TX, typename TC, int SM, int TI>
int launch_up_ti(const void* u, const void* rhs, void* out, const void* e_c, double* partials, const LegGeom& g, bool norm,
                 hipStream_t st) {
  const Coef c = coefs(g.hx, g.hy, g.sigma);
  if (norm) {
    constexpr int HALO = 2 * mg::sweep_halo(SM) + 1;
    mg::FusedArgs a = fused_args<T, HALO, TI>(g.nx, g.ny, g.ld, g.nsweep, !c.pow2, g.nxc, g.nyc, g.ldc, g.poff);
    apply_sub(a, g);
    auto k = g.acoef ? mg::fused_jacobi_kernel<T, HALO, true, mg::kPostNorm, false, TX, TC, 0, SM, TI, true>
           : g.fine ? mg::fused_jacobi_kernel<T, HALO, true, mg::kPostNorm, false, TX, TC, 1, SM, TI>
                    : mg::fused_jacobi_kernel<T, HALO, true, mg::kPostNorm, false, TX, TC, 0, SM, TI>;
    hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(mg::kFusedBlock), 0, st, (const T*)u, (const T*)rhs, (T*)out, (const TX*)e_c,
                       (TX*)nullptr, partials, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD, (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)g.coeff,
                       (const T*)g.acoef, (T)g.sigma, (const T*)g.rdiag);
    return a.ntiles;
  }
  constexpr int HALO = 2 * mg::sweep_halo(SM);
  mg::FusedArgs a = fused_args<T, HALO, TI>(g.nx, g.ny, g.ld, g.nsweep, !c.pow2, g.nxc, g.nyc, g.ldc, g.poff);
  apply_sub(a, g);
  auto k = g.acoef ? mg::fused_jacobi_kernel<T, HALO, true, mg::kPostNone, false, TX, TC, 0, SM, TI, true>
         : g.fine ? mg::fused_jacobi_kernel<T, HALO, true, mg::kPostNone, false, TX, TC, 1, SM, TI>
                  : mg::fused_jacobi_kernel<T, HALO, true, mg::kPostNone, false, TX, TC, 0, SM, TI>;
  hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(mg::kFusedBlock), 0, st, (const T*)u, (const T*)rhs, (T*)out, (const TX*)e_c,
                     (TX*)nullptr, (double*)nullptr, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD, (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)g.coeff,
                     (const T*)g.acoef, (T)g.sigma, (const T*)g.rdiag);
  return 0;
}
template <typename T, typename TX, typename TC, int SM>
int launch_up(const void* u, const void* rhs, void* out, const void* e_c, double* partials, const LegGeom& g, bool norm,
              hipStream_t st) {
  if (tiny_tiles(g)) return launch_up_ti<T, TX, TC, SM, mg::kFusedTITiny>(u, rhs, out, e_c, partials, g, norm, st);
  return small_tiles(g, SM) ? launch_up_ti<T, TX, TC, SM, mg::kFusedTISmall>(u, rhs, out, e_c, partials, g, norm, st)
                        : launch_up_ti<T, TX, TC, SM, mg::kFusedTI>(u, rhs, out, e_c, partials, g, norm, st);
}

// plain multi-sweep smoothing (nsweep <= 2 per launch)
template <typename T, int SM, int TI>
void launch_sweeps_ti(const void* u, const void* rhs, void* out, const LegGeom& g, hipStream_t st) {
  constexpr int HALO = 2 * mg::sweep_halo(SM);
  const Coef c = coefs(g.hx, g.hy, g.sigma);
  const mg::FusedArgs a = fused_args<T, HALO, TI>(g.nx, g.ny, g.ld, g.nsweep, !c.pow2, 0, 0, 0, g.poff);
  auto k = g.acoef ? mg::fused_jacobi_kernel<T, HALO, false, mg::kPostNone, false, T, T, 0, SM, TI, true>
         : g.fine ? mg::fused_jacobi_kernel<T, HALO, false, mg::kPostNone, false, T, T, 1, SM, TI>
                  : mg::fused_jacobi_kernel<T, HALO, false, mg::kPostNone, false, T, T, 0, SM, TI>;
  hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(mg::kFusedBlock), 0, st, (const T*)u, (const T*)rhs, (T*)out, (const T*)nullptr,
                     (T*)nullptr, (double*)nullptr, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD, (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)0,
                     (const T*)g.acoef, (T)g.sigma, (const T*)g.rdiag);
}
template <typename T, int SM>
void launch_sweeps(const void* u, const void* rhs, void* out, const LegGeom& g, hipStream_t st) {
  if (tiny_tiles(g)) launch_sweeps_ti<T, SM, mg::kFusedTITiny>(u, rhs, out, g, st);
  else if (small_tiles(g, SM)) launch_sweeps_ti<T, SM, mg::kFusedTISmall>(u, rhs, out, g, st);
  else launch_sweeps_ti<T, SM, mg::kFusedTI>(u, rhs, out, g, st);
}

// ---- register-blocked legs (mg_rb_kernels.hpp): constant coefficients, levels above ~1100^2 cells ------------------
template <typename T, int HALO, int W, int RPT>
mg::FusedArgs rb_args(const LegGeom& g, bool use_div) {
  using S = mg::RbShape<T, HALO, W, RPT>;
  mg::FusedArgs a = fused_args<T, HALO, mg::kFusedTI>(g.nx, g.ny, g.ld, g.nsweep, use_div, g.nxc, g.nyc, g.ldc, g.poff);
  const int tiles_i = (g.nx - 2 + S::TI - 1) / S::TI;
  a.tiles_j = (g.ny - 1 + S::TJ - 1) / S::TJ;
  a.ntiles = tiles_i * a.tiles_j;
  apply_sub(a, g);
  return a;
}
// g.rb: 0 never, 1 on levels above ~1100^2 cells (where a launch is bandwidth-bound), 2 on every level (tests)
inline bool use_rb(const LegGeom& g, int) { return g.rb == 2 || (g.rb == 1 && (long long)g.nx * g.ny > 1100LL * 1100LL); }

// Arrays of more than ~100 MB cannot stay in the 256 MiB Infinity Cache from one leg to the next (u, t and rhs compete):
// their legs run with streaming hints (rb_leg_kernel TAG 2).  MG_RB_NT=0/1 overrides (experiments).
inline bool rb_stream(const LegGeom& g, size_t esz) {
  static const int force = exp_env("MG_RB_NT", -1);
  if (force >= 0) return force != 0;
  return (size_t)g.nx * g.ld * esz > (size_t)100 << 20;
}
// kernel variant by streaming hints (TAG 2) -- VAR is a template argument of the launcher so that only the shapes in use
// are instantiated: constant coefficients 4 waves x 8 rows (fp64 red-black GS, whose halo is 6 rows: 8 x 8, a 64-row
// region of which 52 rows are tile instead of 20 of 32 -- down / up leg 95 / 98 -> 91 / 91 us at 4097^2; the other three
// lose 3-30 % with it), variable coefficients 8 waves x 4 rows (a 32-row region; the face means of 4 rows per lane keep
// the kernel at ~125 VGPRs instead of 256; 16 x 4 measured 192 / 194 us against 165 / 180)
// measurement builds (-DMG_EXPERIMENTS -DMG_EXP_VAR_W=16 / -DMG_EXP_RB_W=8): other workgroup shapes of the same kernels
#if !defined(MG_EXPERIMENTS) && (defined(MG_EXP_VAR_W) || defined(MG_EXP_RB_W))
#error "MG_EXP_* switches need -DMG_EXPERIMENTS (a measurement build, never the shipped library)"
#endif
#ifndef MG_EXP_VAR_W
#define MG_EXP_VAR_W 8
#endif
#ifndef MG_EXP_RB_W
#define MG_EXP_RB_W ((SM == mg::kSmRbgs && sizeof(T) == 8) ? 8 : 4)
#endif
#define MG_RB_PICK(nt, ...) ((nt) ? mg::rb_leg_kernel<__VA_ARGS__, 2, SM, W, RPT, VAR> : mg::rb_leg_kernel<__VA_ARGS__, 1, SM, W, RPT, VAR>)
template <typename T, typename TX, int SM, int W, int RPT, bool VAR>
void launch_down_rb_s(const void* u, const void* rhs, void* out, void* rhs_c, const LegGeom& g, bool zero_init, hipStream_t st) {
  constexpr int HALO = 2 * mg::sweep_halo(SM) + 2;
  const Coef c = coefs(g.hx, g.hy, g.sigma);
  const mg::FusedArgs a = rb_args<T, HALO, W, RPT>(g, !c.pow2);
  const bool nt = rb_stream(g, sizeof(T));
  auto k = zero_init ? MG_RB_PICK(nt, T, HALO, false, mg::kPostRestrict, true, TX, T)
                     : MG_RB_PICK(nt, T, HALO, false, mg::kPostRestrict, false, TX, T);
  hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(W * 64), 0, st, (const T*)u, (const T*)rhs, (T*)out, (const TX*)nullptr, (TX*)rhs_c,
                     (double*)nullptr, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD, (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)g.coeff,
                     (const T*)g.acoef, (T)g.sigma, (const T*)g.rdiag);
}
template <typename T, typename TX, typename TC, int SM, int W, int RPT, bool VAR>
int launch_up_rb_s(const void* u, const void* rhs, void* out, const void* e_c, double* partials, const LegGeom& g, bool norm, hipStream_t st) {
  const Coef c = coefs(g.hx, g.hy, g.sigma);
  const bool nt = rb_stream(g, sizeof(T));
  if (norm) {
    constexpr int HALO = 2 * mg::sweep_halo(SM) + 1;
    const mg::FusedArgs a = rb_args<T, HALO, W, RPT>(g, !c.pow2);
    auto k = MG_RB_PICK(nt, T, HALO, true, mg::kPostNorm, false, TX, TC);
    hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(W * 64), 0, st,
                       (const T*)u, (const T*)rhs, (T*)out, (const TX*)e_c, (TX*)nullptr, partials, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD,
                       (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)g.coeff, (const T*)g.acoef, (T)g.sigma, (const T*)g.rdiag);
    return a.ntiles;
  }
  constexpr int HALO = 2 * mg::sweep_halo(SM);
  const mg::FusedArgs a = rb_args<T, HALO, W, RPT>(g, !c.pow2);
  auto k = MG_RB_PICK(nt, T, HALO, true, mg::kPostNone, false, TX, TC);
  hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(W * 64), 0, st,
                     (const T*)u, (const T*)rhs, (T*)out, (const TX*)e_c, (TX*)nullptr, (double*)nullptr, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD,
                     (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)g.coeff, (const T*)g.acoef, (T)g.sigma, (const T*)g.rdiag);
  return 0;
}
template <typename T, int SM, int W, int RPT, bool VAR>
void launch_sweeps_rb_s(const void* u, const void* rhs, void* out, const LegGeom& g, hipStream_t st) {
  constexpr int HALO = 2 * mg::sweep_halo(SM);
  const Coef c = coefs(g.hx, g.hy, g.sigma);
  const mg::FusedArgs a = rb_args<T, HALO, W, RPT>(g, !c.pow2);
  const bool nt = rb_stream(g, sizeof(T));
  auto k = MG_RB_PICK(nt, T, HALO, false, mg::kPostNone, false, T, T);
  hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(W * 64), 0, st,
                     (const T*)u, (const T*)rhs, (T*)out, (const T*)nullptr, (T*)nullptr, (double*)nullptr, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD,
                     (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)0, (const T*)g.acoef, (T)g.sigma, (const T*)g.rdiag);
}
template <typename T, typename TX, int SM>
void launch_down_rb(const void* u, const void* rhs, void* out, void* rhs_c, const LegGeom& g, bool zero_init, hipStream_t st) {
  if (g.acoef) launch_down_rb_s<T, TX, SM, MG_EXP_VAR_W, 4, true>(u, rhs, out, rhs_c, g, zero_init, st);
  else launch_down_rb_s<T, TX, SM, MG_EXP_RB_W, 8, false>(u, rhs, out, rhs_c, g, zero_init, st);
}
template <typename T, typename TX, typename TC, int SM>
int launch_up_rb(const void* u, const void* rhs, void* out, const void* e_c, double* partials, const LegGeom& g, bool norm, hipStream_t st) {
  return g.acoef ? launch_up_rb_s<T, TX, TC, SM, MG_EXP_VAR_W, 4, true>(u, rhs, out, e_c, partials, g, norm, st)
                 : launch_up_rb_s<T, TX, TC, SM, MG_EXP_RB_W, 8, false>(u, rhs, out, e_c, partials, g, norm, st);
}
template <typename T, int SM>
void launch_sweeps_rb(const void* u, const void* rhs, void* out, const LegGeom& g, hipStream_t st) {
  if (g.acoef) launch_sweeps_rb_s<T, SM, MG_EXP_VAR_W, 4, true>(u, rhs, out, g, st);
  else launch_sweeps_rb_s<T, SM, MG_EXP_RB_W, 8, false>(u, rhs, out, g, st);
}

// Spanning leg (rb_span_kernel): 8 waves x 8 rows -- the halo of two sweep sets + residual + restriction is 6 rows
// (Jacobi), 52 of the region's 64 rows are tile.  Returns the number of norm partials.
template <typename T, typename TX, typename TC, int SM>
int launch_span_rb(const void* u, const void* rhs, void* out_mid, void* out_next, const void* e_c, void* rhs_c, double* partials,
                   const LegGeom& g, int nsweep_pre, hipStream_t st) {
  constexpr int W = 8, RPT = 8;
  constexpr int HALO = 4 * mg::sweep_halo(SM) + 2;
  const Coef c = coefs(g.hx, g.hy, g.sigma);
  mg::FusedArgs a = rb_args<T, HALO, W, RPT>(g, !c.pow2);
  a.nsweep2 = nsweep_pre;
  static const int band = std::max(1, exp_env("MG_EXP_SPAN_BAND", 4));      // measurement builds: other band heights
  a.band = band;
  const bool nt = rb_stream(g, sizeof(T));
  auto k = out_mid ? (nt ? mg::rb_span_kernel<T, HALO, TX, TC, 2, SM, W, RPT, 1> : mg::rb_span_kernel<T, HALO, TX, TC, 1, SM, W, RPT, 1>)
                   : (nt ? mg::rb_span_kernel<T, HALO, TX, TC, 2, SM, W, RPT, 2> : mg::rb_span_kernel<T, HALO, TX, TC, 1, SM, W, RPT, 2>);
  hipLaunchKernelGGL(k, dim3(a.ntiles), dim3(W * 64), 0, st, (const T*)u, (const T*)rhs, (T*)out_mid, (T*)out_next, (const TX*)e_c,
                     (TX*)rhs_c, partials, a, (T)c.ihx2, (T)c.ihy2, (T)c.invD, (T)c.diag, (T)g.omega, (T)(1.0 - g.omega), (T)g.coeff);
  return a.ntiles;
}
// dt: dtype of the level and of the level below, dcomp: interpolation dtype.  -1: no spanning leg for this combination.
template <int SM>
int d_span_sm(int dt, int dcomp, const void* u, const void* rhs, void* out_mid, void* out_next, const void* e_c, void* rhs_c,
              double* partials, const LegGeom& g, int nsweep_pre, hipStream_t st) {
  if (dt == MG_F64 && dcomp == MG_F64) return launch_span_rb<double, double, double, SM>(u, rhs, out_mid, out_next, e_c, rhs_c, partials, g, nsweep_pre, st);
  if (dt == MG_F32 && dcomp == MG_F64) return launch_span_rb<float, float, double, SM>(u, rhs, out_mid, out_next, e_c, rhs_c, partials, g, nsweep_pre, st);
  if (dt == MG_F32 && dcomp == MG_F32) return launch_span_rb<float, float, float, SM>(u, rhs, out_mid, out_next, e_c, rhs_c, partials, g, nsweep_pre, st);
  return -1;
}
int d_span(int dt, int dcomp, const void* u, const void* rhs, void* out_mid, void* out_next, const void* e_c, void* rhs_c,
           double* partials, const LegGeom& g, int nsweep_pre, hipStream_t st, int sm = MG_JACOBI) {
  return sm == MG_RBGS ? d_span_sm<mg::kSmRbgs>(dt, dcomp, u, rhs, out_mid, out_next, e_c, rhs_c, partials, g, nsweep_pre, st)
                       : d_span_sm<mg::kSmJacobi>(dt, dcomp, u, rhs, out_mid, out_next, e_c, rhs_c, partials, g, nsweep_pre, st);
}

// One weighted-Jacobi sweep on a level above ~1100^2 cells: the register-blocked sweeps kernel with nsweep = 1 (same
// arithmetic, same ping-pong contract as jacobi_kernel: interior rows written, the ring of `out` already equals u's).
// MG_JACOBI_RB=0 keeps the LDS-tiled jacobi_kernel (A/B runs).
bool jacobi_rb(int dt, const void* u, const void* rhs, void* out, int nx, int ny, int ld, double hx, double hy, double omega,
               hipStream_t st, double sigma) {
  static const int on = exp_env("MG_JACOBI_RB", 1);
  LegGeom g{nx, ny, ld, 0, 0, 0, hx, hy, omega, 0.0, 1, 0, true};
  g.sigma = sigma; g.rb = 1;
  // only where the arrays stream from HBM (4097^2 fp64: 81 -> 78 us); Infinity-Cache-resident sweeps are faster LDS-tiled
  if (!on || !use_rb(g, mg::kSmJacobi) || !rb_stream(g, esize(dt))) return false;
  if (dt == MG_F32) launch_sweeps_rb<float, mg::kSmJacobi>(u, rhs, out, g, st); else launch_sweeps_rb<double, mg::kSmJacobi>(u, rhs, out, g, st);
  return true;
}

template <int SM>
void d_down_sm(int dt, int dx, const void* u, const void* rhs, void* out, void* rhs_c, const LegGeom& g, bool zero_init, hipStream_t st) {
  if (use_rb(g, SM)) {
    if (dt == MG_F32 && dx == MG_F32) launch_down_rb<float, float, SM>(u, rhs, out, rhs_c, g, zero_init, st);
    else if (dt == MG_F64 && dx == MG_F64) launch_down_rb<double, double, SM>(u, rhs, out, rhs_c, g, zero_init, st);
    else if (dt == MG_F64 && dx == MG_F32) launch_down_rb<double, float, SM>(u, rhs, out, rhs_c, g, zero_init, st);
    else launch_down_rb<float, double, SM>(u, rhs, out, rhs_c, g, zero_init, st);
    return;
  }
  if (dt == MG_F32 && dx == MG_F32) launch_down<float, float, SM>(u, rhs, out, rhs_c, g, zero_init, st);
  else if (dt == MG_F64 && dx == MG_F64) launch_down<double, double, SM>(u, rhs, out, rhs_c, g, zero_init, st);
  else if (dt == MG_F64 && dx == MG_F32) launch_down<double, float, SM>(u, rhs, out, rhs_c, g, zero_init, st);
  else launch_down<float, double, SM>(u, rhs, out, rhs_c, g, zero_init, st);
}
void d_down(int sm, int dt, int dx, const void* u, const void* rhs, void* out, void* rhs_c, const LegGeom& g, bool zero_init,
            hipStream_t st) {
  if (sm == MG_RBGS) d_down_sm<mg::kSmRbgs>(dt, dx, u, rhs, out, rhs_c, g, zero_init, st);
  else d_down_sm<mg::kSmJacobi>(dt, dx, u, rhs, out, rhs_c, g, zero_init, st);
}
// dt: fine dtype, dx: coarse e dtype, dcomp: interpolation dtype.  Returns #partials, or -1 for an unsupported combination.
template <int SM>
int d_up_sm(int dt, int dx, int dcomp, const void* u, const void* rhs, void* out, const void* e_c, double* partials,
            const LegGeom& g, bool norm, hipStream_t st) {
  if (use_rb(g, SM)) {
    if (dcomp == MG_F32) {
      if (dt == MG_F32 && dx == MG_F32) return launch_up_rb<float, float, float, SM>(u, rhs, out, e_c, partials, g, norm, st);
      return -1;
    }
    if (dt == MG_F64 && dx == MG_F64) return launch_up_rb<double, double, double, SM>(u, rhs, out, e_c, partials, g, norm, st);
    if (dt == MG_F64 && dx == MG_F32) return launch_up_rb<double, float, double, SM>(u, rhs, out, e_c, partials, g, norm, st);
    if (dt == MG_F32 && dx == MG_F64) return launch_up_rb<float, double, double, SM>(u, rhs, out, e_c, partials, g, norm, st);
    return launch_up_rb<float, float, double, SM>(u, rhs, out, e_c, partials, g, norm, st);
  }
  if (dcomp == MG_F32) {
    if (dt == MG_F32 && dx == MG_F32) return launch_up<float, float, float, SM>(u, rhs, out, e_c, partials, g, norm, st);
    return -1;
  }
  if (dt == MG_F64 && dx == MG_F64) return launch_up<double, double, double, SM>(u, rhs, out, e_c, partials, g, norm, st);
  if (dt == MG_F64 && dx == MG_F32) return launch_up<double, float, double, SM>(u, rhs, out, e_c, partials, g, norm, st);
  if (dt == MG_F32 && dx == MG_F64) return launch_up<float, double, double, SM>(u, rhs, out, e_c, partials, g, norm, st);
  return launch_up<float, float, double, SM>(u, rhs, out, e_c, partials, g, norm, st);
}
int d_up(int sm, int dt, int dx, int dcomp, const void* u, const void* rhs, void* out, const void* e_c, double* partials,
         const LegGeom& g, bool norm, hipStream_t st) {
  return sm == MG_RBGS ? d_up_sm<mg::kSmRbgs>(dt, dx, dcomp, u, rhs, out, e_c, partials, g, norm, st)
                       : d_up_sm<mg::kSmJacobi>(dt, dx, dcomp, u, rhs, out, e_c, partials, g, norm, st);
}
void d_sweeps(int sm, int dt, const void* u, const void* rhs, void* out, const LegGeom& g, hipStream_t st) {
  if (use_rb(g, sm == MG_RBGS ? mg::kSmRbgs : mg::kSmJacobi)) {
    if (sm == MG_RBGS) { if (dt == MG_F32) launch_sweeps_rb<float, mg::kSmRbgs>(u, rhs, out, g, st); else launch_sweeps_rb<double, mg::kSmRbgs>(u, rhs, out, g, st); }
    else { if (dt == MG_F32) launch_sweeps_rb<float, mg::kSmJacobi>(u, rhs, out, g, st); else launch_sweeps_rb<double, mg::kSmJacobi>(u, rhs, out, g, st); }
    return;
  }
  if (sm == MG_RBGS) {
    if (dt == MG_F32) launch_sweeps<float, mg::kSmRbgs>(u, rhs, out, g, st); else launch_sweeps<double, mg::kSmRbgs>(u, rhs, out, g, st);
  } else {
    if (dt == MG_F32) launch_sweeps<float, mg::kSmJacobi>(u, rhs, out, g, st); else launch_sweeps<double, mg::kSmJacobi>(u, rhs, out, g, st);
  }
}

template <typename TI, typename TO>
void launch_inject_ring(const void* fine, void* coarse, int nxf, int nyf, int ldf, int nxc, int nyc, int ldc, int sides, int ci_off,
                        int cj_off, hipStream_t st) {
  hipLaunchKernelGGL((mg::inject_ring_kernel<TI, TO>), dim3(grid_for(2 * (nxc + nyc))), dim3(mg::kBlock), 0, st,
                     (const TI*)fine, (TO*)coarse, nxf, nyf, ldf, nxc, nyc, ldc, sides, ci_off, cj_off);
}

void d_inject_ring(int di, int dout, const void* fine, void* coarse, int nxf, int nyf, int ldf, int nxc, int nyc, int ldc,
                   hipStream_t st, int sides = mg::kAllSides, int ci_off = 0, int cj_off = 0) {
  if (di == MG_F32 && dout == MG_F32) launch_inject_ring<float, float>(fine, coarse, nxf, nyf, ldf, nxc, nyc, ldc, sides, ci_off, cj_off, st);
  else if (di == MG_F64 && dout == MG_F64) launch_inject_ring<double, double>(fine, coarse, nxf, nyf, ldf, nxc, nyc, ldc, sides, ci_off, cj_off, st);
  else if (di == MG_F64 && dout == MG_F32) launch_inject_ring<double, float>(fine, coarse, nxf, nyf, ldf, nxc, nyc, ldc, sides, ci_off, cj_off, st);
  else launch_inject_ring<float, double>(fine, coarse, nxf, nyf, ldf, nxc, nyc, ldc, sides, ci_off, cj_off, st);
}

// ------------------------------------------------------------------ host <-> device helpers -----
int upload(std::string* err, void* dev, int ddt, int ld, const void* host, int hdt, int nx, int ny, void* staging,
           hipStream_t st) {
  if (ddt == hdt) {
    HIPC(err, hipMemcpy2DAsync(dev, (size_t)ld * esize(ddt), host, (size_t)ny * esize(hdt), (size_t)ny * esize(hdt), nx,
                               hipMemcpyHostToDevice, st));
  } else {   // upload in the host dtype into staging (pitch = ld of the HOST dtype), then cast on the device
    const int lds = pitch_elems(hdt, ny);
    HIPC(err, hipMemcpy2DAsync(staging, (size_t)lds * esize(hdt), host, (size_t)ny * esize(hdt), (size_t)ny * esize(hdt),
                               nx, hipMemcpyHostToDevice, st));
    d_convert(hdt, ddt, staging, dev, nx, ny, lds, ld, st);
  }
  HIPC(err, hipStreamSynchronize(st));
  return MG_OK;
}

int download(std::string* err, void* host, int hdt, const void* dev, int ddt, int ld, int nx, int ny, void* staging,
             hipStream_t st) {
  if (ddt == hdt) {
    HIPC(err, hipMemcpy2DAsync(host, (size_t)ny * esize(hdt), dev, (size_t)ld * esize(ddt), (size_t)ny * esize(hdt), nx,
                               hipMemcpyDeviceToHost, st));
  } else {
    const int lds = pitch_elems(hdt, ny);
    d_convert(ddt, hdt, dev, staging, nx, ny, ld, lds, st);
    HIPC(err, hipMemcpy2DAsync(host, (size_t)ny * esize(hdt), staging, (size_t)lds * esize(hdt), (size_t)ny * esize(hdt),
                               nx, hipMemcpyDeviceToHost, st));
  }
  HIPC(err, hipStreamSynchronize(st));
  return MG_OK;
}

}  // namespace

namespace {

// cfg.fused: 0 one launch per operator, 1 LDS-tiled fused legs, 2 register-blocked legs on the large levels (LDS-tiled
// below), 3 register-blocked legs on every level
inline int rb_mode(const mg_handle* h) { return h->cfg.fused == 2 ? 1 : (h->cfg.fused == 3 ? 2 : 0); }

// hipMemset on device memory is asynchronous to the host and runs on the NULL stream, which a
// hipStreamNonBlocking stream does not wait for: zero on the stream that will use the memory.
int alloc_zero(std::string* err, void** p, size_t bytes, hipStream_t st = nullptr) {
  HIPC(err, hipMalloc(p, bytes));
  HIPC(err, hipMemsetAsync(*p, 0, bytes, st));
  return MG_OK;
}

void release(mg_handle* h) {
  for (auto& l : h->lv)
    for (int d = 0; d < 2; ++d) {
      if (l.u[d]) (void)hipFree(l.u[d]);
      if (l.t[d]) (void)hipFree(l.t[d]);
      if (l.s[d]) (void)hipFree(l.s[d]);
      if (l.rhs[d]) (void)hipFree(l.rhs[d]);
      if (l.r[d]) (void)hipFree(l.r[d]);
      if (l.a[d]) (void)hipFree(l.a[d]);
      if (l.rd[d]) (void)hipFree(l.rd[d]);
    }
  if (h->d_tail_ops) (void)hipFree(h->d_tail_ops);
  if (h->partials) (void)hipFree(h->partials);
  if (h->d_scalar) (void)hipFree(h->d_scalar);
  if (h->d_int) (void)hipFree(h->d_int);
  if (h->d_minv) (void)hipFree(h->d_minv);
  if (h->staging) (void)hipFree(h->staging);
  if (h->mbox) (void)hipHostFree(h->mbox);
  if (h->h_scalar) (void)hipHostFree(h->h_scalar);
  if (h->h_int) (void)hipHostFree(h->h_int);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
}

// reciprocal diagonals of every level and precision, for the current coefficient and shift (variable coefficients only)
void refresh_rdiag(mg_handle* h) {
  if (!h->varcoef || h->rd_sigma == h->sigma) return;
  for (auto& v : h->lv)
    for (int dt = 0; dt < 2; ++dt)
      if (v.a[dt] && v.rd[dt]) d_rdiag(dt, v.a[dt], v.rd[dt], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->sigma, h->stream);
  h->rd_sigma = h->sigma;
}

// ---- the cycle -----------------------------------------------------------------------------
struct StageTimer {
  mg_handle* h; Level* lv; int slot; double t0 = 0;
  StageTimer(mg_handle* h_, Level* lv_, int slot_) : h(h_), lv(lv_), slot(slot_) {
    if (h->cfg.profile) { (void)hipStreamSynchronize(h->stream); t0 = now_s(); }
  }
  ~StageTimer() {
    if (h->cfg.profile) { (void)hipStreamSynchronize(h->stream); lv->timings[slot] += now_s() - t0; }
  }
};

void smooth(mg_handle* h, int l, int nu) {
  Level& v = h->lv[l];
  const int dt = h->level_dtype(l);
  StageTimer tm(h, &v, 0);
  for (int s = 0; s < nu; ++s) {
    if (h->varcoef && h->cfg.smoother == MG_JACOBI) {
      d_var<mg::kVarJacobi>(dt, v.u[dt], v.a[dt], v.rhs[dt], v.t[dt], nullptr, v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.omega,
                            h->cfg.coeff, 0, 0, h->stream, h->sigma);
      std::swap(v.u[dt], v.t[dt]);
    } else if (h->varcoef && h->cfg.smoother == MG_RBGS) {
      for (int colour = 0; colour < 2; ++colour)
        d_var<mg::kVarRbgs>(dt, v.u[dt], v.a[dt], v.rhs[dt], v.u[dt], nullptr, v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.omega,
                            h->cfg.coeff, colour, h->cfg.colour_offset, h->stream, h->sigma);
    } else if (h->cfg.smoother == MG_JACOBI) {
      d_jacobi(dt, v.u[dt], v.rhs[dt], v.t[dt], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.omega, h->stream, l == 0, h->sigma);
      std::swap(v.u[dt], v.t[dt]);
    } else if (h->cfg.smoother == MG_RBGS) {
      for (int colour = 0; colour < 2; ++colour)
        d_rbgs_colour(dt, v.u[dt], v.rhs[dt], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.omega, colour,
                      h->cfg.colour_offset, h->stream, l == 0, h->sigma);
    } else {   // MG_LEXGS: exactly `nu` sweeps (tol < 0 never triggers the early exit)
      d_coarse(dt, v.u[dt], v.rhs[dt], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.coeff, h->cfg.omega, -1.0, nu - s,
               nullptr, h->stream, false, h->varcoef ? v.a[dt] : nullptr, h->sigma);
      break;
    }
  }
}

void coarse_solve(mg_handle* h, int l, bool zero_init = false) {
  Level& v = h->lv[l];
  const int dt = h->level_dtype(l);
  // solvers/multigrid.py:119-124: the default coarse solver is GaussSeidelSmoother(omega = 1)
  d_coarse(dt, v.u[dt], v.rhs[dt], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.coeff, 1.0, h->cfg.coarse_tol,
           h->cfg.coarse_maxit, h->d_int, h->stream, zero_init, h->varcoef ? v.a[dt] : nullptr, h->sigma);
}

int cycle(mg_handle* h, int l) {
  const int L = h->L();
  if (l == L - 1) { coarse_solve(h, l); return MG_OK; }
  Level& f = h->lv[l];
  Level& c = h->lv[l + 1];
  const int dt = h->level_dtype(l), dc = h->level_dtype(l + 1);
  if (h->cfg.pre > 0) smooth(h, l, h->cfg.pre);
  {
    StageTimer tm(h, &f, 1);
    if (h->varcoef)
      d_var<mg::kVarResidual>(dt, f.u[dt], f.a[dt], f.rhs[dt], f.r[dt], nullptr, f.nx, f.ny, f.ld[dt], f.hx, f.hy, 1.0,
                              h->cfg.coeff, 0, 0, h->stream, h->sigma);
    else
      d_residual(dt, f.u[dt], f.rhs[dt], f.r[dt], f.nx, f.ny, f.ld[dt], f.hx, f.hy, h->cfg.coeff, h->stream, l == 0, h->sigma);
    d_restrict(dt, dc, f.r[dt], c.rhs[dc], f.nx, f.ny, f.ld[dt], c.ld[dc], h->stream);
  }
  (void)hipMemsetAsync(c.u[dc], 0, (size_t)c.nx * c.ld[dc] * esize(dc), h->stream);
  int reps = 1;
  if (h->cfg.cycle == MG_CYCLE_W) reps = 2;
  else if (h->cfg.cycle == MG_CYCLE_F) reps = std::max(1, 1 << std::max(0, L - l - 2));   // multigrid.py:315-319
  for (int k = 0; k < reps; ++k) {
    const int rc = cycle(h, l + 1);
    if (rc != MG_OK) return rc;
  }
  {
    StageTimer tm(h, &f, 2);
    const int rc = d_prolong<true>(dc, dt, h->grid_dtype, c.u[dc], f.u[dt], f.nx, f.ny, f.ld[dt], c.ld[dc], h->stream);
    if (rc != MG_OK) return rc;
  }
  if (h->cfg.post > 0) smooth(h, l, h->cfg.post);
  return MG_OK;
}


// ------------------------------------------------------------------ coarse tail (one workgroup, LDS) ----
constexpr size_t kTailPoolLimit = 150 * 1024;

size_t tail_pool_bytes(const mg_handle* h, int k, size_t esz, size_t esz_last) {
  size_t b = 0;
  const size_t extra = h->varcoef ? 1 : 0;          // the coefficient field of every level rides along
  for (int l = k; l < h->L(); ++l) {
    const size_t cells = (size_t)h->lv[l].nx * h->lv[l].ny;
    b += (l == h->L() - 1) ? (2 + extra) * cells * esz_last : (3 + extra) * cells * esz;
    b = (b + 15) / 16 * 16;
  }
  return b + (size_t)mg::kPipeCells * mg::kPipeSlots * esz_last;   // snapshot ring of the pipelined coarsest solve
}

void tail_schedule(const mg_handle* h, int k, int l, int zero_flag, std::vector<int>& ops) {
  const int L = h->L();
  if (l == L - 1) { ops.push_back(mg::kTailSolve | ((l - k) << 8) | (zero_flag << 16)); return; }
  ops.push_back(mg::kTailDown | ((l - k) << 8) | (zero_flag << 16));
  int reps = 1;
  if (h->cfg.cycle == MG_CYCLE_W) reps = 2;
  else if (h->cfg.cycle == MG_CYCLE_F) reps = std::max(1, 1 << std::max(0, L - l - 2));
  for (int r = 0; r < reps; ++r) tail_schedule(h, k, l + 1, r == 0 ? 1 : 0, ops);
  ops.push_back(mg::kTailUp | ((l - k) << 8));
}

template <typename T, typename TCO, typename TC>
int tail_set_attr(size_t bytes) {
  return (hipFuncSetAttribute(reinterpret_cast<const void*>(&mg::coarse_tail_kernel<T, TCO, TC, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(&mg::coarse_tail_kernel<T, TCO, TC, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) ? MG_OK : MG_ERR_HIP;
}

// mg_config.coarse_direct: 1 or < 0 (the host side's default) the nine unknowns of a 5 x 5 coarsest grid are solved directly,
// 0 by the reference's iteration to coarse_tol (bit-identical to it; what the parity tests pin).  W- and F-cycles visit the
// coarsest level 2^(L-1) times per cycle and spent most of their time in that iteration; a V-cycle saves its ~20 sweeps.
bool want_direct(const mg_handle* h) {
  const int L = h->L();
  const int n = (h->lv[L - 1].nx - 2) * (h->lv[L - 1].ny - 2);          // unknowns of the coarsest grid: 9 for the 5 x 5 of every
  if (n < 1 || n > 64) return false;                                    // 2^k + 1 square, 21 for the 9 x 5 of a 2:1 domain
  return h->cfg.coarse_direct != 0;
}

// Decide where the tail starts: the first level k >= 1 whose sub-hierarchy fits the LDS pool, has at most
// kTailMaxLevels levels and (per-level MIXED policy) one dtype on levels k .. L-2.
int plan_tail_lds(mg_handle* h);
int plan_tail(mg_handle* h) {
  int rc = plan_tail_lds(h);
  if (rc != MG_OK) return rc;
  rc = tail2_plan(h);                       // the register-resident tail takes precedence where it applies
  if (rc == MG_OK && h->tail2_start >= 0) h->tail_direct = want_direct(h);
  return rc;
}
int plan_tail_lds(mg_handle* h) {
  h->tail_start = -1;
  h->tail2_start = -1;
  h->tail_direct = false;
  if (h->d_tail_ops) { (void)hipFree(h->d_tail_ops); h->d_tail_ops = nullptr; }      // re-planned when the operator changes
  const int L = h->L();
  if (!h->fused() || L < 3 || h->cfg.pre > 8 || h->cfg.post > 8) return MG_OK;
  const size_t esz_last = esize(h->grid_dtype);
  static const int max_top = exp_env("MG_EXP_TAIL_MAXN", 0);   // experiment: largest top level
  for (int k = 1; k <= L - 2; ++k) {
    if (L - k > mg::kTailMaxLevels) continue;
    const size_t esz = (h->cfg.precision == MG_PREC_ADAPTIVE) ? 8 : esize(h->level_dtype_in(k, MG_F64));
    if (tail_pool_bytes(h, k, esz, esz_last) > kTailPoolLimit) continue;
    // A 65^2 fp64 level costs more as two stages of the one-workgroup tail (12 us: five LDS reads per cell and stage through
    // one CU's LDS) than as two launches of its own (2 x 4.8 us): fp64 tails start at 33^2 (bench step -8 us, W(2,2) -3 %,
    // config 1 / 2 -10 / -6 %; profiles/README.md).  fp32 levels move half the bytes and stay in the tail from 65^2.
    const int cap = max_top > 0 ? max_top : (esz == 8 ? 33 : 65);
    if (std::max(h->lv[k].nx, h->lv[k].ny) > cap) continue;
    bool uniform = true;
    for (int l = k; l <= L - 2; ++l) uniform = uniform && (h->level_dtype_in(l, MG_F64) == h->level_dtype_in(k, MG_F64));
    if (!uniform) continue;
    h->tail_start = k;
    break;
  }
  h->tail_direct = false;
  h->tail_minv_sigma = -1.0;
  if (h->tail_start < 0) return MG_OK;
  h->tail_direct = want_direct(h);
  std::vector<int> ops;
  tail_schedule(h, h->tail_start, h->tail_start, 2, ops);
  h->tail_nops = (int)ops.size();
  HIPC(&h->err, hipMalloc((void**)&h->d_tail_ops, sizeof(int) * ops.size()));
  HIPC(&h->err, hipMemcpy(h->d_tail_ops, ops.data(), sizeof(int) * ops.size(), hipMemcpyHostToDevice));
  const size_t lim = kTailPoolLimit + 1024;
  if (tail_set_attr<double, double, double>(lim) != MG_OK || tail_set_attr<float, float, float>(lim) != MG_OK ||
      tail_set_attr<float, double, double>(lim) != MG_OK) { h->tail_start = -1; }
  return MG_OK;
}

// Inverse of the coarsest 5 x 5 system the smoothers relax: (-div(a grad .) + sigma) u = f on the nine interior cells,
// zero ring (a == 1 without a coefficient field).  Gaussian elimination with partial pivoting in long double.
int build_coarse_inverse(mg_handle* h) {
  const Level& v = h->lv[h->L() - 1];
  const int cnx = v.nx, cny = v.ny, my = cny - 2, n = (cnx - 2) * my;
  if (n < 1 || n > 64) return fail(&h->err, MG_ERR_INVALID_VALUE, "coarse_direct: the coarsest grid has more than 64 unknowns");
  std::vector<double> a((size_t)cnx * cny, 1.0);
  if (h->varcoef) {
    const int dt = h->grid_dtype;
    std::vector<unsigned char> buf((size_t)cnx * v.ld[dt] * esize(dt));
    HIPC(&h->err, hipMemcpyAsync(buf.data(), v.a[dt], buf.size(), hipMemcpyDeviceToHost, h->stream));
    HIPC(&h->err, hipStreamSynchronize(h->stream));
    for (int i = 0; i < cnx; ++i)
      for (int j = 0; j < cny; ++j)
        a[(size_t)i * cny + j] = dt == MG_F32 ? (double)reinterpret_cast<const float*>(buf.data())[(size_t)i * v.ld[dt] + j]
                                              : reinterpret_cast<const double*>(buf.data())[(size_t)i * v.ld[dt] + j];
  }
  const long double ihx2 = 1.0L / ((long double)v.hx * v.hx), ihy2 = 1.0L / ((long double)v.hy * v.hy);
  const int w = 2 * n;
  std::vector<long double> M((size_t)n * w, 0.0L);      // [A | I], unknown (i, j) at p = (i - 1) my + (j - 1)
  for (int p = 0; p < n; ++p) {
    M[(size_t)p * w + n + p] = 1.0L;
    const int i = p / my + 1, j = p % my + 1;
    const long double c = a[(size_t)i * cny + j];
    const long double aip = 0.5L * (c + a[(size_t)(i + 1) * cny + j]), aim = 0.5L * (c + a[(size_t)(i - 1) * cny + j]);
    const long double ajp = 0.5L * (c + a[(size_t)i * cny + j + 1]), ajm = 0.5L * (c + a[(size_t)i * cny + j - 1]);
    M[(size_t)p * w + p] = (aip + aim) * ihx2 + (ajp + ajm) * ihy2 + (long double)h->sigma;
    if (i < cnx - 2) M[(size_t)p * w + p + my] = -aip * ihx2;
    if (i > 1) M[(size_t)p * w + p - my] = -aim * ihx2;
    if (j < my) M[(size_t)p * w + p + 1] = -ajp * ihy2;
    if (j > 1) M[(size_t)p * w + p - 1] = -ajm * ihy2;
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r) if (fabsl(M[(size_t)r * w + c]) > fabsl(M[(size_t)piv * w + c])) piv = r;
    if (M[(size_t)piv * w + c] == 0.0L) return fail(&h->err, MG_ERR_INVALID_VALUE, "coarse_direct: singular coarsest system");
    if (piv != c) for (int q = 0; q < w; ++q) std::swap(M[(size_t)piv * w + q], M[(size_t)c * w + q]);
    const long double d = 1.0L / M[(size_t)c * w + c];
    for (int q = 0; q < w; ++q) M[(size_t)c * w + q] *= d;
    for (int r = 0; r < n; ++r) {
      if (r == c || M[(size_t)r * w + c] == 0.0L) continue;
      const long double f = M[(size_t)r * w + c];
      for (int q = 0; q < w; ++q) M[(size_t)r * w + q] -= f * M[(size_t)c * w + q];
    }
  }
  if (n == 9) {
    for (int p = 0; p < 9; ++p)
      for (int q = 0; q < 9; ++q) h->tail_minv[p * 9 + q] = (double)M[(size_t)p * w + 9 + q];
  }
  if (cnx != 5 || cny != 5) {      // the LDS tail streams the rows of the inverse from device memory
    std::vector<double> inv((size_t)n * n);
    for (int p = 0; p < n; ++p)
      for (int q = 0; q < n; ++q) inv[(size_t)p * n + q] = (double)M[(size_t)p * w + n + q];
    if (!h->d_minv) HIPC(&h->err, hipMalloc((void**)&h->d_minv, sizeof(double) * 64 * 64));
    HIPC(&h->err, hipStreamSynchronize(h->stream));                  // no launch in flight reads the old inverse
    HIPC(&h->err, hipMemcpy(h->d_minv, inv.data(), sizeof(double) * inv.size(), hipMemcpyHostToDevice));
    h->minv_n = n;
  }
  h->tail_minv_sigma = h->sigma;
  return MG_OK;
}

int launch_tail(mg_handle* h, bool zero_top) {
  const int k = h->tail_start, L = h->L();
  const int dt = h->level_dtype(k), dco = h->grid_dtype;
  mg::TailArgs a;
  std::memset(&a, 0, sizeof(a));
  a.nlev = L - k; a.nops = h->tail_nops; a.pre = h->cfg.pre; a.post = h->cfg.post;
  a.ld_top = h->lv[k].ld[dt]; a.maxit = h->cfg.coarse_maxit;
  a.omega = h->cfg.omega; a.coeff = h->cfg.coeff; a.tol = h->cfg.coarse_tol; a.tol_x = sqrt_threshold(h->cfg.coarse_tol);
  a.smoother = (h->cfg.smoother == MG_RBGS) ? mg::kSmRbgs : mg::kSmJacobi; a.colour_offset = h->cfg.colour_offset & 1;
  a.sigma = h->sigma;
  a.direct = 0;
  if (h->tail_direct) {
    if (h->tail_minv_sigma != h->sigma) { const int rc = build_coarse_inverse(h); if (rc != MG_OK) return rc; }
    a.direct = 1;
    std::memcpy(a.minv, h->tail_minv, sizeof(a.minv));
    const Level& cl = h->lv[L - 1];
    if (cl.nx != 5 || cl.ny != 5) { a.minv_dev = h->d_minv; a.minv_n = h->minv_n; }
  }
  const bool var = h->varcoef;
  const size_t extra = var ? 1 : 0;
  size_t off = 0;
  for (int l = k; l < L; ++l) {
    const Level& v = h->lv[l];
    mg::TailLevel& t = a.lv[l - k];
    const Coef c = coefs(v.hx, v.hy, h->sigma);
    t.nx = v.nx; t.ny = v.ny; t.off = (int)off;
    t.ihx2 = c.ihx2; t.ihy2 = c.ihy2; t.invD = c.invD; t.diag = c.diag; t.hx2 = v.hx * v.hx; t.hy2 = v.hy * v.hy;
    t.hxhy = v.hx * v.hy; t.use_div = c.pow2 ? 0 : 1; t.exact_recip = c.all_pow2 ? 1 : 0;
    if (var) {     // hx^2, hy^2 powers of two are all the variable-coefficient solve needs to multiply by reciprocals
      int e = 0;
      t.exact_recip = (std::frexp(t.hx2, &e) == 0.5 && std::frexp(t.hy2, &e) == 0.5) ? 1 : 0;
      const int dl = (l == L - 1) ? dco : dt;
      a.a_lv[l - k] = v.a[dl]; a.a_ld[l - k] = v.ld[dl];
    }
    const size_t cells = (size_t)v.nx * v.ny;
    off += (l == L - 1) ? (2 + extra) * cells * esize(dco) : (3 + extra) * cells * esize(dt);
    off = (off + 15) / 16 * 16;
  }
  off += (size_t)mg::kPipeCells * mg::kPipeSlots * esize(dco);
  Level& top = h->lv[k];
  const dim3 grid(1), block(mg::kTailBlock);
#define MG_TAIL_LAUNCH(T, TCO, TC, VAR)                                                                                     \
  hipLaunchKernelGGL((mg::coarse_tail_kernel<T, TCO, TC, VAR>), grid, block, off, h->stream, (const T*)top.rhs[dt], (T*)top.u[dt], \
                     h->d_tail_ops, a, zero_top ? 1 : 0, h->d_int)
  if (dt == MG_F64) { if (var) MG_TAIL_LAUNCH(double, double, double, true); else MG_TAIL_LAUNCH(double, double, double, false); }
  else if (dco == MG_F32) { if (var) MG_TAIL_LAUNCH(float, float, float, true); else MG_TAIL_LAUNCH(float, float, float, false); }
  else { if (var) MG_TAIL_LAUNCH(float, double, double, true); else MG_TAIL_LAUNCH(float, double, double, false); }
#undef MG_TAIL_LAUNCH
  return MG_OK;
}

// Fused V/W/F-cycle: two launches per level (down leg, up leg) instead of nine.  Same arithmetic per cell.
// zero_u: the iterate of this level is the zero correction and need not be read (first visit of a coarse level).
constexpr int kPartFull = 0, kPartFront = 1, kPartBack = 2;

// `part` (level 0 only) splits the cycle for speculative launching: the FRONT part (down leg + the whole
// sub-cycle below) never writes the buffer that holds the current fine iterate, only the BACK part (up leg) does.
int cycle_fused(mg_handle* h, int l, bool zero_u, int part = kPartFull) {
  const int L = h->L();
  Level& f = h->lv[l];
  const int dt = h->level_dtype(l);
  const size_t bytes = (size_t)f.nx * f.ld[dt] * esize(dt);
  if (l == h->tail2_start && h->cfg.tail != 0) {
    if (h->tail_direct && h->tail_minv_sigma != h->sigma) { const int rc = build_coarse_inverse(h); if (rc != MG_OK) return rc; }
    return tail2_launch(h, zero_u);
  }
  // the LDS tail only where the register-resident one does not apply (a variable-coefficient fp32 hierarchy could enter the
  // LDS tail one level earlier, at 65^2: 107 us per launch against two 33^2 register tails and one pair of legs, ~70)
  if (l == h->tail_start && h->cfg.tail != 0 && h->tail2_start < 0) return launch_tail(h, zero_u);
  if (l == L - 1) {
    coarse_solve(h, l, zero_u);
    return MG_OK;
  }
  Level& c = h->lv[l + 1];
  const int dc = h->level_dtype(l + 1);
  const bool fine = (l == 0);
  const int sm = h->cfg.smoother;
  LegGeom g{f.nx, f.ny, f.ld[dt], c.nx, c.ny, c.ld[dc], f.hx, f.hy, h->cfg.omega, h->cfg.coeff, 0, h->cfg.colour_offset, fine};
  g.sigma = h->sigma;
  g.rb = rb_mode(h);
  if (h->varcoef) { g.acoef = f.a[dt]; g.rdiag = f.rd[dt]; }
  if (part != kPartBack) {
    StageTimer tm(h, &f, 0);
    int extra = std::max(0, h->cfg.pre - 2);
    if (extra > 0 && zero_u) { (void)hipMemsetAsync(f.u[dt], 0, bytes, h->stream); zero_u = false; }
    while (extra > 0) {
      g.nsweep = std::min(2, extra);
      d_sweeps(sm, dt, f.u[dt], f.rhs[dt], f.t[dt], g, h->stream);
      std::swap(f.u[dt], f.t[dt]);
      extra -= g.nsweep;
    }
    g.nsweep = std::min(2, h->cfg.pre);
    d_down(sm, dt, dc, f.u[dt], f.rhs[dt], f.t[dt], c.rhs[dc], g, zero_u, h->stream);
    std::swap(f.u[dt], f.t[dt]);
  }
  int reps = 1;
  if (h->cfg.cycle == MG_CYCLE_W) reps = 2;
  else if (h->cfg.cycle == MG_CYCLE_F) reps = std::max(1, 1 << std::max(0, L - l - 2));
  for (int k = 0; k < reps && part != kPartBack; ++k) {
    const int rc = cycle_fused(h, l + 1, k == 0);
    if (rc != MG_OK) return rc;
  }
  if (part != kPartFront) {
    StageTimer tm(h, &f, 2);
    const bool want_norm = fine && h->cfg.post <= 2 && h->cfg.precision != MG_PREC_DEFECT;   // defect correction: the norm is the fp64 defect's
    g.nsweep = std::min(2, h->cfg.post);
    const int n = d_up(sm, dt, dc, h->grid_dtype, f.u[dt], f.rhs[dt], f.t[dt], c.u[dc], h->partials, g, want_norm, h->stream);
    if (n < 0) return MG_ERR_INVALID_VALUE;
    std::swap(f.u[dt], f.t[dt]);
    if (fine) h->norm_partials = want_norm ? n : 0;
    int extra = std::max(0, h->cfg.post - 2);
    while (extra > 0) {
      g.nsweep = std::min(2, extra);
      d_sweeps(sm, dt, f.u[dt], f.rhs[dt], f.t[dt], g, h->stream);
      std::swap(f.u[dt], f.t[dt]);
      extra -= g.nsweep;
    }
  }
  return MG_OK;
}

// Spanning leg: the BACK part of the running cycle (level-0 up leg) and the FRONT part of the next one (level-0 down leg,
// then everything below) with ONE level-0 launch in place of two.  Buffers: u holds the pre-smoothed iterate of the running
// cycle; the kernel writes the iterate of that cycle to t (keep_mid; dropped when the caller knows the solve cannot end
// there) and the pre-smoothed iterate of the next cycle to the third buffer s.  Afterwards u = s (what the next up leg
// reads), t = the iterate (undo_front's swap brings it back), s = the buffer just consumed.
bool span_ok(const mg_handle* h) {
  if (h->cfg.speculate < 2 || !h->fused() || h->L() < 3 || h->varcoef) return false;
  if (h->cfg.pre < 1 || h->cfg.pre > 2 || h->cfg.post < 1 || h->cfg.post > 2 || h->cfg.precision == MG_PREC_DEFECT) return false;
  const Level& f = h->lv[0];
  LegGeom g{f.nx, f.ny, 0, 0, 0, 0, f.hx, f.hy, 0, 0, 0, 0, true};
  g.rb = rb_mode(h);
  if (!use_rb(g, mg::kSmJacobi)) return false;                       // bandwidth-bound levels only
  return h->level_dtype(0) == h->level_dtype(1) && (h->level_dtype(0) == MG_F32 || h->grid_dtype == MG_F64);
}
int cycle_span(mg_handle* h, bool keep_mid) {
  const int L = h->L();
  Level& f = h->lv[0];
  Level& c = h->lv[1];
  const int dt = h->level_dtype(0), dc = h->level_dtype(1);
  if (!f.s[dt]) {
    const int rc = alloc_zero(&h->err, &f.s[dt], (size_t)f.nx * f.ld[dt] * esize(dt), h->stream);
    if (rc != MG_OK) return rc;
    h->span_ring[dt] = false;
  }
  if (!h->span_ring[dt]) {                                           // the Dirichlet ring, once per solve and working precision
    d_convert_ring(dt, dt, f.u[dt], f.s[dt], f.nx, f.ny, f.ld[dt], f.ld[dt], h->stream);
    h->span_ring[dt] = true;
  }
  LegGeom g{f.nx, f.ny, f.ld[dt], c.nx, c.ny, c.ld[dc], f.hx, f.hy, h->cfg.omega, h->cfg.coeff, 0, h->cfg.colour_offset, true};
  g.sigma = h->sigma;
  g.rb = rb_mode(h);
  g.nsweep = h->cfg.post;
  const int n = d_span(dt, h->grid_dtype, f.u[dt], f.rhs[dt], keep_mid ? f.t[dt] : nullptr, f.s[dt], c.u[dc], c.rhs[dc], h->partials, g,
                       h->cfg.pre, h->stream, h->cfg.smoother);
  if (n < 0) return MG_ERR_INVALID_VALUE;
  void* consumed = f.u[dt];
  f.u[dt] = f.s[dt];
  f.s[dt] = consumed;
  h->norm_partials = n;
  return MG_OK;
}
// ... and the rest of that front part: the sub-cycle(s) below level 0 (queued after the norm reduction of the cycle before)
int cycle_below_fine(mg_handle* h) {
  const int L = h->L();
  int reps = 1;
  if (h->cfg.cycle == MG_CYCLE_W) reps = 2;
  else if (h->cfg.cycle == MG_CYCLE_F) reps = std::max(1, 1 << std::max(0, L - 2));
  for (int k = 0; k < reps; ++k) {
    const int rc = cycle_fused(h, 1, k == 0);
    if (rc != MG_OK) return rc;
  }
  return MG_OK;
}

// Full-multigrid initial guess (solvers/advanced_multigrid.py:626-683, gpu/gpu_solver.py:603-652): restrict the rhs to
// every level (full weighting), solve the coarsest level from zero, then walk up: u_l = P u_{l+1}, followed by
// `ncyc` cycles of the sub-hierarchy that starts at level l.  The boundary ring of the fine iterate (Dirichlet data)
// is kept; coarser rings are zero.
int fmg_init(mg_handle* h, int ncyc) {
  h->iterate_zero = false;
  const int L = h->L();
  if (L < 2) return MG_OK;
  h->norm_partials = 0;
  for (int l = 0; l + 1 < L; ++l) {
    Level& f = h->lv[l];
    Level& c = h->lv[l + 1];
    const int dt = h->level_dtype(l), dc = h->level_dtype(l + 1);
    d_restrict(dt, dc, f.rhs[dt], c.rhs[dc], f.nx, f.ny, f.ld[dt], c.ld[dc], h->stream);
  }
  {
    Level& v = h->lv[L - 1];
    const int dt = h->level_dtype(L - 1);
    (void)hipMemsetAsync(v.u[dt], 0, (size_t)v.nx * v.ld[dt] * esize(dt), h->stream);
    coarse_solve(h, L - 1, false);
  }
  for (int l = L - 2; l >= 0; --l) {
    Level& f = h->lv[l];
    Level& c = h->lv[l + 1];
    const int dt = h->level_dtype(l), dc = h->level_dtype(l + 1);
    const size_t bytes = (size_t)f.nx * f.ld[dt] * esize(dt);
    if (l > 0) {
      (void)hipMemsetAsync(f.u[dt], 0, bytes, h->stream);
      if (f.t[dt]) (void)hipMemsetAsync(f.t[dt], 0, bytes, h->stream);
    } else {
      // keep the Dirichlet ring, zero the interior (whatever an initial guess or earlier cycles left there); the
      // ping-pong partner t carries the same ring by invariant and is fully rewritten by the first leg
      d_zero_interior(dt, f.u[dt], f.nx, f.ny, f.ld[dt], h->stream);
    }
    // interior cells: u = P e  (ADD = false writes every cell the interpolation defines, ring included: e ring is 0)
    if (l > 0) {
      if (d_prolong<false>(dc, dt, h->grid_dtype, c.u[dc], f.u[dt], f.nx, f.ny, f.ld[dt], c.ld[dc], h->stream) != MG_OK) return MG_ERR_INVALID_VALUE;
    } else {
      // level 0: u = ring + P e on the interior == (ring-only field) + P e everywhere, because (P e)[ring] = 0
      if (d_prolong<true>(dc, dt, h->grid_dtype, c.u[dc], f.u[dt], f.nx, f.ny, f.ld[dt], c.ld[dc], h->stream) != MG_OK) return MG_ERR_INVALID_VALUE;
    }
    for (int k = 0; k < ncyc; ++k) {
      const int rc = (h->fused()) ? cycle_fused(h, l, false) : cycle(h, l);
      if (rc != MG_OK) return rc;
    }
  }
  h->norm_partials = 0;
  HIPC(&h->err, hipGetLastError());
  return MG_OK;
}

void inject_rings_fwd(mg_handle* h);      // = inject_rings(h, h->phase), defined below
int launch_defect(mg_handle* h, bool update);

// Full-multigrid start under defect correction (MG_PREC_DEFECT): the fp32 hierarchy solves the ERROR equation, so the FMG
// pass runs on A e = f - A u0 (u0: the Dirichlet ring of the fp64 iterate, zero inside) from the zero correction, and the
// result is added to the fp64 iterate -- the defect loop then starts from u0 + e instead of discarding the FMG work.
int defect_fmg(mg_handle* h, int ncyc) {
  if (h->varcoef) return MG_ERR_INVALID_VALUE;
  if (h->L() < 2) return MG_OK;
  Level& v = h->lv[0];
  (void)launch_defect(h, false);                     // rhs[fp32] = f - A u (zero ring)
  inject_rings_fwd(h);                               // (zero) rings of every coarse rhs
  const size_t bytes = (size_t)v.nx * v.ld[MG_F32] * 4;
  (void)hipMemsetAsync(v.u[MG_F32], 0, bytes, h->stream);
  if (v.t[MG_F32]) (void)hipMemsetAsync(v.t[MG_F32], 0, bytes, h->stream);
  const int rc = fmg_init(h, ncyc);                  // e in lv[0].u[fp32]
  if (rc != MG_OK) return rc;
  (void)launch_defect(h, true);                      // u <- u + e in fp64 (and the next defect)
  h->norm_partials = 0;
  h->iterate_zero = false;
  return MG_OK;
}

// ---- defect correction (MG_PREC_DEFECT): fp64 iterate and residual, fp32 cycles on the error equation -----------------
// One pass over the fine grid per outer step: u <- u + e (the fp32 correction of the cycle just run), r = f - A u in
// fp64, stored as the fp32 right-hand side of the next error equation (zero on boundary cells), sum r^2 for the norm.
// Returns the number of partials.
int launch_defect(mg_handle* h, bool update) {
  Level& v = h->lv[0];
  const Coef c = coefs(v.hx, v.hy, h->sigma);
  const long long pairs = (long long)v.nx * ((v.ny + 1) / 2);
  const int nb = std::min(grid_for(pairs), 2048);
  if (update) {
    hipLaunchKernelGGL((mg::residual_xprec_kernel<double, double, float, true, true, true>), dim3(nb), dim3(mg::kBlock), 0, h->stream,
                       (const double*)v.u[MG_F64], (const float*)v.u[MG_F32], (const double*)v.rhs[MG_F64], (float*)v.rhs[MG_F32],
                       (double*)v.t[MG_F64], h->partials, v.nx, v.ny, v.ld[MG_F64], v.ld[MG_F32], v.ld[MG_F64], v.ld[MG_F32], c.ihx2,
                       c.ihy2, c.diag, h->cfg.coeff);
    std::swap(v.u[MG_F64], v.t[MG_F64]);
  } else {
    hipLaunchKernelGGL((mg::residual_xprec_kernel<double, double, float, false, true, true>), dim3(nb), dim3(mg::kBlock), 0, h->stream,
                       (const double*)v.u[MG_F64], (const float*)nullptr, (const double*)v.rhs[MG_F64], (float*)v.rhs[MG_F32],
                       (double*)nullptr, h->partials, v.nx, v.ny, v.ld[MG_F64], v.ld[MG_F32], v.ld[MG_F64], v.ld[MG_F32], c.ihx2,
                       c.ihy2, c.diag, h->cfg.coeff);
  }
  return nb;
}

// one fp32 cycle from the zero correction on the current defect (left in lv[0].u[fp32])
int defect_cycle(mg_handle* h) {
  h->norm_partials = 0;
  h->iterate_zero = false;
  if (h->L() == 1) {                     // a single level: the "cycle" is the coarsest solve, in the grid dtype (fp64)
    return MG_ERR_INVALID_VALUE;
  }
  if (h->fused()) return cycle_fused(h, 0, true);
  Level& v = h->lv[0];
  (void)hipMemsetAsync(v.u[MG_F32], 0, (size_t)v.nx * v.ld[MG_F32] * 4, h->stream);
  return cycle(h, 0);
}

int run_cycle(mg_handle* h) {
  h->norm_partials = 0;
  h->iterate_zero = false;
  if (h->cfg.precision == MG_PREC_DEFECT) {            // one outer step: defect, fp32 cycle, update
    if (h->varcoef) return MG_ERR_INVALID_VALUE;
    (void)launch_defect(h, false);
    inject_rings_fwd(h);
    const int rc = defect_cycle(h);
    if (rc != MG_OK) return rc;
    (void)launch_defect(h, true);
    return MG_OK;
  }
  if (h->fused() && h->L() > 1) return cycle_fused(h, 0, false);
  return cycle(h, 0);
}

// Reduce `n` partial sums and bring the scalar to the host.  Fast path: the kernel posts it to the mapped
// mailbox and the host spins on the sequence number (a few microseconds after the kernel retires); if nothing
// arrives within 2 s, or there is no mailbox, fall back to copy + stream synchronisation.
// launch half of reduce_to_host (mailbox only): returns the sequence number to wait for
unsigned long long reduce_post(mg_handle* h, int n) {
  const unsigned long long seq = ++h->mbox_seq;
  launch_reduce(h->partials, n, h->d_scalar, h->stream, h->mbox_dev, seq);
  return seq;
}

int reduce_wait(mg_handle* h, unsigned long long seq, double* value) {
  volatile unsigned long long* flag = &h->mbox->seq;
  const double t0 = now_s();
  long spins = 0;
  while (*flag != seq) {
    if ((++spins & 0x3fff) == 0 && now_s() - t0 > 2.0) break;
  }
  if (*flag != seq) HIPC(&h->err, hipStreamSynchronize(h->stream));     // slow or faulted device
  if (*flag != seq) return fail(&h->err, MG_ERR_HIP, "norm mailbox was never written");
  std::atomic_thread_fence(std::memory_order_acquire);
  *value = *(volatile double*)&h->mbox->value;
  return MG_OK;
}

int reduce_to_host(mg_handle* h, int n, double* value) {
  if (h->mbox_dev) {
    const unsigned long long seq = ++h->mbox_seq;
    launch_reduce(h->partials, n, h->d_scalar, h->stream, h->mbox_dev, seq);
    HIPC(&h->err, hipGetLastError());
    volatile unsigned long long* flag = &h->mbox->seq;
    const double t0 = now_s();
    long spins = 0;
    while (*flag != seq) {
      if ((++spins & 0x3fff) == 0 && now_s() - t0 > 2.0) break;
    }
    if (*flag == seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      *value = *(volatile double*)&h->mbox->value;
      return MG_OK;
    }
    HIPC(&h->err, hipStreamSynchronize(h->stream));     // slow or faulted device: this reports the error, if any
    if (*flag == seq) { *value = *(volatile double*)&h->mbox->value; return MG_OK; }
  } else {
    launch_reduce(h->partials, n, h->d_scalar, h->stream);
  }
  HIPC(&h->err, hipMemcpyAsync(h->h_scalar, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPC(&h->err, hipStreamSynchronize(h->stream));
  *value = *h->h_scalar;
  return MG_OK;
}

int fine_norm(mg_handle* h, double* out) {
  Level& v = h->lv[0];
  if (h->cfg.precision == MG_PREC_DEFECT) {       // ||f - A u|| of the fp64 iterate (the fp32 rhs is rewritten with the same defect)
    if (h->varcoef) return fail(&h->err, MG_ERR_INVALID_VALUE, "defect correction runs the constant-coefficient operator");
    const int n = launch_defect(h, false);
    double ss = 0;
    const int rc = reduce_to_host(h, n, &ss);
    if (rc != MG_OK) return rc;
    *out = std::sqrt(v.hx * v.hy * ss);
    return MG_OK;
  }
  const int dt = h->level_dtype(0);
  if (h->norm_partials > 0 && h->ring_sumsq[dt] >= 0) {   // the up leg of the last cycle already summed r^2 over the interior cells
    double ss = 0;
    const int rc = reduce_to_host(h, h->norm_partials, &ss);
    if (rc != MG_OK) return rc;
    *out = std::sqrt(v.hx * v.hy * (ss + h->ring_sumsq[dt]));
    return MG_OK;
  }
  // the norm of the zero iterate is ||f||, whatever the operator: computed once per right-hand side (by the same kernel, so
  // with the same bits) and remembered -- repeated solves of one right-hand side from the zero guess skip the pass
  if (h->iterate_zero && h->zero_norm_gen[dt] == h->rhs_gen) { *out = h->zero_norm_val[dt]; return MG_OK; }
  const int n = h->varcoef
      ? d_var<mg::kVarResidualNorm>(dt, v.u[dt], v.a[dt], v.rhs[dt], nullptr, h->partials, v.nx, v.ny, v.ld[dt], v.hx, v.hy,
                                    1.0, h->cfg.coeff, 0, 0, h->stream, h->sigma)
      : d_residual_norm(dt, v.u[dt], v.rhs[dt], h->partials, v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.coeff, h->stream, true,
                        h->sigma);
  double ss = 0;
  const int rc = reduce_to_host(h, n, &ss);
  if (rc != MG_OK) return rc;
  *out = std::sqrt(v.hx * v.hy * ss);
  if (h->iterate_zero) { h->zero_norm_gen[dt] = h->rhs_gen; h->zero_norm_val[dt] = *out; }
  return MG_OK;
}

void inject_rings(mg_handle* h, int ph, bool only_shared = false);

// in-device cast of the fine iterate when the adaptive policy changes the working precision
// iterate_is_zero: the iterate is zero everywhere and the caller starts the coming cycle from the zero iterate without
// reading it (cycle_fused(..., zero_u = true)): only the (zero) boundary rings of the two buffers are written.
int switch_phase(mg_handle* h, int to, bool iterate_is_zero = false) {
  if (h->cfg.precision != MG_PREC_ADAPTIVE || to == h->phase) return MG_OK;
  Level& v = h->lv[0];
  const int from = h->phase;
  if (h->L() > 1) {   // with a single level the only level is the coarsest and lives in the grid dtype
    if (iterate_is_zero) d_convert_ring(from, to, v.u[from], v.u[to], v.nx, v.ny, v.ld[from], v.ld[to], h->stream);
    else d_convert(from, to, v.u[from], v.u[to], v.nx, v.ny, v.ld[from], v.ld[to], h->stream);
    // the ping-pong partner only needs the boundary ring (the first leg rewrites its interior)
    if (v.t[to]) d_convert_ring(from, to, v.u[from], v.t[to], v.nx, v.ny, v.ld[from], v.ld[to], h->stream);
  }
  h->phase = to;
  h->norm_partials = 0;
  if (h->have_rhs) {     // once per right-hand side and working precision, except the arrays the two precisions share
    inject_rings(h, to, h->rings_gen[to] == h->rhs_gen);
    h->rings_gen[to] = h->rhs_gen;
  }
  return MG_OK;
}

// core/precision.py:189-246 should_promote_precision, on the last five residual norms
bool stagnating(const std::vector<double>& hist) {
  if (hist.size() < 5) return false;
  const double* r = hist.data() + hist.size() - 5;
  double sum = 0; int n = 0;
  for (int i = 1; i < 5; ++i) if (r[i - 1] > 0) { sum += r[i] / r[i - 1]; ++n; }
  if (n) {
    if (sum / n > 0.9) return true;
    double rel = 0; int m = 0;
    for (int i = 1; i < 5; ++i) if (r[i - 1] > 0) { rel += std::fabs(r[i] - r[i - 1]) / r[i - 1]; ++m; }
    if (m && rel / m < 1e-3) return true;
  }
  bool inc = true;
  for (int i = 1; i < 5; ++i) inc = inc && (r[i] >= r[i - 1] * 0.99);
  return inc;
}

// The residual an fp32 iterate can reach: every cell's r = f - A u carries the rounding of diag(A) u, so ||r||_h settles at
// about eps32 * diag(A) * ||u||_h (measured: 1.8 against 2.0 at 4097^2, 1.9e-3 against 1.95e-3 at 129^2) -- h^-2 times the
// round-off of u.  Within a factor 2 of it another fp32 cycle cannot lower the residual: the policy promotes at once
// instead of waiting for the five-norm stagnation window (core/precision.py:189-246) to fill with a flat history.
bool at_fp32_floor(const mg_handle* h, double rn) { return h->fp32_floor > 0.0 && rn <= 2.0 * h->fp32_floor; }

// Before the first cycle the same floor can be bounded from above: ||u|| <= ||f|| / lambda_min with lambda_min =
// |coeff| pi^2 (1/Lx^2 + 1/Ly^2) + sigma of the Dirichlet problem, so floor / ||r_0|| <= eps32 diag(A) / lambda_min -- a number
// that depends on the grid only (1.2e-8 / h^2 on the unit square: 0.2 at 4097^2, 0.013 at 1025^2).  Cycles contract ||r|| by
// ~0.15, so the fp32 phase is good for log(that) / log(0.15) cycles; entering and leaving it costs about one cycle's saving
// (two ring conversions, the ||u|| pass, the cast of the iterate, no speculative front part across either switch): the policy
// takes the fp32 phase only when it is good for at least two cycles, and stays in double otherwise -- an adaptive solve then
// never loses to a double one.
bool fp32_phase_pays(const mg_handle* h) {
  const Level& v = h->lv[0];
  const double lx = h->cfg.x1 - h->cfg.x0, ly = h->cfg.y1 - h->cfg.y0, pi = 3.14159265358979323846;
  const double lam = std::fabs(h->cfg.coeff) * pi * pi * (1.0 / (lx * lx) + 1.0 / (ly * ly)) + h->sigma;
  const double ratio = 5.9604644775390625e-8 * coefs(v.hx, v.hy, h->sigma).diag / lam;
  return ratio > 0.0 && std::log(ratio) / std::log(0.15) >= 2.0;
}

// core/precision.py:270-302 update_precision (+ the one-way variant documented in mghip.h): the precision the coming
// cycle runs in.  Pure -- `*promote` says whether taking the decision also ends the adaptive phase for good.
int adapt_target(const mg_handle* h, double rn, bool* promote) {
  *promote = false;
  if (h->cfg.precision != MG_PREC_ADAPTIVE) return h->phase;
  const double thr = h->cfg.switch_threshold;
  double pts = 0;
  for (auto& l : h->lv) pts += (double)l.nx * l.ny;
  const double mem = pts * esize(h->phase) * 4.0;                       // precision.py:136-153
  const bool mem_down = mem > h->cfg.memory_threshold_gb * 1024.0 * 1024.0 * 1024.0;
  int to = h->phase;
  if (h->cfg.adaptive_reference_rule) {
    if (mem_down || (h->phase == MG_F64 && rn > thr * 100)) { if (h->phase == MG_F64) to = MG_F32; }
    else if (h->phase == MG_F32 && rn < thr * 10) to = MG_F64;
  } else if (!h->promoted) {
    if (h->phase == MG_F64 && (mem_down || (rn > thr * 100 && fp32_phase_pays(h))) && h->adapt_hist.empty()) to = MG_F32;
    else if (h->phase == MG_F32 && (rn < thr * 10 || stagnating(h->adapt_hist) || at_fp32_floor(h, rn))) { to = MG_F64; *promote = true; }
  }
  return to;
}

int adapt(mg_handle* h, double rn, bool iterate_is_zero = false) {
  bool promote = false;
  const int to = adapt_target(h, rn, &promote);
  if (promote) {
    h->promoted = true;
    h->switch_reason = rn < h->cfg.switch_threshold * 10 ? 1 : (stagnating(h->adapt_hist) ? 2 : 3);
  } else if (h->cfg.precision == MG_PREC_ADAPTIVE && !h->cfg.adaptive_reference_rule && !h->promoted && h->phase == MG_F64 &&
             to == MG_F64 && h->adapt_hist.empty() && rn > h->cfg.switch_threshold * 100) {
    h->switch_reason = 4;                                  // the fp32 phase was declined a priori (fp32_phase_pays)
    h->promoted = true;                                    // ... for good: the rest of the solve is a double solve
  }
  return switch_phase(h, to, iterate_is_zero);
}

// The boundary ring of every coarse rhs is the injected fine ring (r = f on boundary cells, injection on the
// coarse boundary: operators/laplacian.py:117-118, operators/transfer.py:109-113): constant over a solve, so it
// is written here once per rhs (and per working precision) instead of in every cycle.
// only_shared: just the coarse arrays that BOTH working precisions of the adaptive policy use (the fp64 coarsest level: its
// ring is the injected ring of an fp32 rhs in one phase and of the fp64 rhs in the other); the rest is still valid.
void inject_rings(mg_handle* h, int ph, bool only_shared) {
  for (int l = 0; l + 1 < h->L(); ++l) {
    Level& f = h->lv[l];
    Level& c = h->lv[l + 1];
    const int dt = h->level_dtype_in(l, ph), dc = h->level_dtype_in(l + 1, ph);
    if (only_shared && h->level_dtype_in(l + 1, MG_F32) != h->level_dtype_in(l + 1, MG_F64)) continue;
    d_inject_ring(dt, dc, f.rhs[dt], c.rhs[dc], f.nx, f.ny, f.ld[dt], c.nx, c.ny, c.ld[dc], h->stream);
  }
}

void inject_rings_fwd(mg_handle* h) { inject_rings(h, h->phase); }

// sum of f^2 over the boundary ring of the fine rhs (4 windows of the device reduction), per allocated dtype
int ring_sums(mg_handle* h) {
  Level& v = h->lv[0];
  for (int dt = 0; dt < 2; ++dt) {
    h->ring_sumsq[dt] = 0;
    if (!v.rhs[dt]) continue;
    const int win[4][4] = {{0, 1, 0, v.ny}, {v.nx - 1, v.nx, 0, v.ny}, {1, v.nx - 1, 0, 1}, {1, v.nx - 1, v.ny - 1, v.ny}};
    for (int k = 0; k < 4; ++k) {
      const int n = d_sumsq(dt, v.rhs[dt], h->partials, v.ld[dt], win[k][0], win[k][1], win[k][2], win[k][3], h->stream);
      launch_reduce(h->partials, n, h->d_scalar, h->stream);
      HIPC(&h->err, hipMemcpyAsync(h->h_scalar, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPC(&h->err, hipStreamSynchronize(h->stream));
      h->ring_sumsq[dt] += *h->h_scalar;
    }
  }
  return MG_OK;
}

int rhs_changed(mg_handle* h) {
  h->have_rhs = true;
  h->norm_partials = 0;
  ++h->rhs_gen;
  inject_rings(h, h->phase);
  h->rings_gen[h->phase & 1] = h->rhs_gen;
  return ring_sums(h);
}

int set_rhs_impl(mg_handle* h, const void* rhs, int hdt) {
  Level& v = h->lv[0];
  for (int dt = 0; dt < 2; ++dt)
    if (v.rhs[dt]) {
      const int rc = upload(&h->err, v.rhs[dt], dt, v.ld[dt], rhs, hdt, v.nx, v.ny, h->staging, h->stream);
      if (rc != MG_OK) return rc;
    }
  return rhs_changed(h);
}

int set_u_impl(mg_handle* h, const void* u0, int hdt) {
  Level& v = h->lv[0];
  // Adaptive policy: every solve starts in double (PrecisionManager's default precision, core/precision.py:26-45), so a
  // new initial guess goes straight into the fp64 iterate instead of being converted up when the solve begins
  if (h->cfg.precision == MG_PREC_ADAPTIVE && h->phase != MG_F64 && h->L() > 1) {
    h->phase = MG_F64;
    if (h->have_rhs) {
      inject_rings(h, MG_F64, h->rings_gen[MG_F64] == h->rhs_gen);
      h->rings_gen[MG_F64] = h->rhs_gen;
    }
  }
  const int dt = h->iterate_dtype();
  h->norm_partials = 0;
  h->iterate_zero = (u0 == nullptr);
  if (u0) {
    int rc = upload(&h->err, v.u[dt], dt, v.ld[dt], u0, hdt, v.nx, v.ny, h->staging, h->stream);
    if (rc != MG_OK) return rc;
    if (v.t[dt])     // the ping-pong partner must carry the same boundary ring
      d_convert_ring(dt, dt, v.u[dt], v.t[dt], v.nx, v.ny, v.ld[dt], v.ld[dt], h->stream);
  } else {
    HIPC(&h->err, hipMemsetAsync(v.u[dt], 0, (size_t)v.nx * v.ld[dt] * esize(dt), h->stream));
    if (v.t[dt]) HIPC(&h->err, hipMemsetAsync(v.t[dt], 0, (size_t)v.nx * v.ld[dt] * esize(dt), h->stream));
  }
  HIPC(&h->err, hipStreamSynchronize(h->stream));
  return MG_OK;
}

}  // namespace

// =================================================================== C ABI =====================
extern "C" {

const char* mg_version(void) { return "mghip 0.1 (gfx950)"; }

int mg_device_count(int* count) {
  if (!count) return fail(nullptr, MG_ERR_INVALID_VALUE, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(nullptr, MG_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *count = n;
  return MG_OK;
}

const char* mg_last_error(const mg_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int mg_pitch_elems(int dtype, int ny, int* ld) {
  if (!valid_dtype(dtype) || ny < 1 || !ld) return fail(nullptr, MG_ERR_INVALID_VALUE, "mg_pitch_elems: bad argument");
  *ld = pitch_elems(dtype, ny);
  return MG_OK;
}

int mg_create(const mg_config* cfg, mg_handle** out) {
  if (!cfg || !out) return fail(nullptr, MG_ERR_INVALID_VALUE, "mg_create: NULL argument");
  *out = nullptr;
  if (cfg->nx < 3 || cfg->ny < 3)
    return fail(nullptr, MG_ERR_INVALID_VALUE, "Grid must have at least 3 points in each direction");   // core/grid.py:34-35
  if (cfg->cycle < MG_CYCLE_V || cfg->cycle > MG_CYCLE_F) return fail(nullptr, MG_ERR_INVALID_VALUE, "unknown cycle type");
  if (cfg->smoother < MG_JACOBI || cfg->smoother > MG_LEXGS) return fail(nullptr, MG_ERR_INVALID_VALUE, "unknown smoother");
  if (cfg->precision < MG_PREC_DOUBLE || cfg->precision > MG_PREC_DEFECT) return fail(nullptr, MG_ERR_INVALID_VALUE, "unknown precision policy");
  if (cfg->pre < 0 || cfg->post < 0 || cfg->max_levels < 1 || cfg->coarse_maxit < 1)
    return fail(nullptr, MG_ERR_INVALID_VALUE, "negative sweep count / max_levels < 1 / coarse_maxit < 1");
  if (!(cfg->x1 > cfg->x0) || !(cfg->y1 > cfg->y0)) return fail(nullptr, MG_ERR_INVALID_VALUE, "empty domain");

  int ndev = 0;
  int rc = mg_device_count(&ndev);
  if (rc != MG_OK) return rc;
  if (ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MG_ERR_NO_DEVICE, "no such HIP device");
  HIPC(nullptr, hipSetDevice(cfg->device));

  mg_handle* h = new mg_handle();
  h->cfg = *cfg;
  h->grid_dtype = (cfg->precision == MG_PREC_SINGLE) ? MG_F32 : MG_F64;
  h->phase = MG_F64;
  // hierarchy: solvers/multigrid.py:135-171
  int nx = cfg->nx, ny = cfg->ny;
  for (int level = 0; level < cfg->max_levels; ++level) {
    if (level > 0) {
      if ((nx - 1) % 2 != 0 || (ny - 1) % 2 != 0) break;          // core/grid.py:148-149
      const int cx = (nx - 1) / 2 + 1, cy = (ny - 1) / 2 + 1;
      if (cx < 5 || cy < 5) break;                                 // multigrid.py:158-160
      nx = cx; ny = cy;
    }
    Level l;
    l.nx = nx; l.ny = ny;
    l.hx = (cfg->x1 - cfg->x0) / (nx - 1);
    l.hy = (cfg->y1 - cfg->y0) / (ny - 1);
    l.ld[0] = pitch_elems(MG_F32, ny);
    l.ld[1] = pitch_elems(MG_F64, ny);
    h->lv.push_back(l);
  }
  auto bail = [&](int code) { release(h); std::string m = h->err; delete h; g_last_error = m; return code; };
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return bail(MG_ERR_HIP); }
  h->stream = h->own_stream;
  for (int l = 0; l < h->L(); ++l) {
    Level& v = h->lv[l];
    for (int dt = 0; dt < 2; ++dt) {
      if (!h->needs(l, dt)) continue;
      const size_t bytes = (size_t)v.nx * v.ld[dt] * esize(dt);
      if ((rc = alloc_zero(&h->err, &v.u[dt], bytes, h->stream)) != MG_OK) return bail(rc);
      if ((rc = alloc_zero(&h->err, &v.rhs[dt], bytes, h->stream)) != MG_OK) return bail(rc);
      const bool master = cfg->precision == MG_PREC_DEFECT && l == 0 && dt == MG_F64;     // iterate + partner + f only
      if (l < h->L() - 1 || master) {
        if (!master && (rc = alloc_zero(&h->err, &v.r[dt], bytes, h->stream)) != MG_OK) return bail(rc);
        if ((master || cfg->smoother == MG_JACOBI || h->fused()) && (rc = alloc_zero(&h->err, &v.t[dt], bytes, h->stream)) != MG_OK) return bail(rc);
      }
    }
  }
  {
    const size_t np = max_partials(cfg->nx, cfg->ny);
    if ((rc = alloc_zero(&h->err, (void**)&h->partials, sizeof(double) * np, h->stream)) != MG_OK) return bail(rc);
  }
  if ((rc = alloc_zero(&h->err, (void**)&h->d_scalar, sizeof(double), h->stream)) != MG_OK) return bail(rc);
  if ((rc = alloc_zero(&h->err, (void**)&h->d_int, sizeof(int), h->stream)) != MG_OK) return bail(rc);
  if ((rc = alloc_zero(&h->err, &h->staging, (size_t)cfg->nx * pitch_elems(MG_F64, cfg->ny) * 8, h->stream)) != MG_OK) return bail(rc);
  if (hipHostMalloc((void**)&h->h_scalar, sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_int, sizeof(int)) != hipSuccess) { h->err = "hipHostMalloc failed"; return bail(MG_ERR_ALLOC); }
  if (hipHostMalloc((void**)&h->mbox, sizeof(mg::HostMailbox), hipHostMallocMapped) == hipSuccess &&
      hipHostGetDevicePointer((void**)&h->mbox_dev, h->mbox, 0) == hipSuccess) {
    h->mbox->value = 0; h->mbox->seq = 0;
  } else {
    h->mbox_dev = nullptr;     // no mapped host memory: fall back to copy + stream synchronisation
  }
  if ((rc = plan_tail(h)) != MG_OK) return bail(rc);
  if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "hipStreamSynchronize failed"; return bail(MG_ERR_HIP); }
  *out = h;
  return MG_OK;
}

int mg_destroy(mg_handle* h) {
  if (!h) return MG_OK;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  release(h);
  delete h;
  return MG_OK;
}

int mg_num_levels(const mg_handle* h, int* n) {
  if (!h || !n) return fail(nullptr, MG_ERR_INVALID_VALUE, "NULL argument");
  *n = h->L();
  return MG_OK;
}

int mg_level_shape(const mg_handle* h, int level, int* nx, int* ny) {
  if (!h || !nx || !ny || level < 0 || level >= h->L()) return fail(nullptr, MG_ERR_INVALID_VALUE, "bad level");
  *nx = h->lv[level].nx; *ny = h->lv[level].ny;
  return MG_OK;
}

int mg_level_timings(const mg_handle* h, int level, double out3[3]) {
  if (!h || !out3 || level < 0 || level >= h->L()) return fail(nullptr, MG_ERR_INVALID_VALUE, "bad level");
  for (int k = 0; k < 3; ++k) out3[k] = h->lv[level].timings[k];
  return MG_OK;
}

int mg_get_stream(mg_handle* h, void** stream) {
  if (!h || !stream) return fail(nullptr, MG_ERR_INVALID_VALUE, "NULL argument");
  *stream = (void*)h->stream;
  return MG_OK;
}

int mg_set_stream(mg_handle* h, void* stream, int use_own) {
  if (!h) return fail(nullptr, MG_ERR_INVALID_VALUE, "NULL handle");
  HIPC(&h->err, hipStreamSynchronize(h->stream));
  h->stream = use_own ? h->own_stream : (hipStream_t)stream;
  return MG_OK;
}

int mg_set_rhs_device(mg_handle* h, const void* rhs_dev, int ld, int dtype) {
  if (!h || !rhs_dev || !valid_dtype(dtype) || ld < h->lv[0].ny) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_set_rhs_device: bad argument");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  Level& v = h->lv[0];
  for (int dt = 0; dt < 2; ++dt)
    if (v.rhs[dt]) d_convert(dtype, dt, rhs_dev, v.rhs[dt], v.nx, v.ny, ld, v.ld[dt], h->stream);
  h->have_rhs = true;
  h->norm_partials = 0;
  ++h->rhs_gen;
  inject_rings(h, h->phase);        // asynchronous; the ring sum (host round trip) is only needed by mg_residual_norm
  h->rings_gen[h->phase & 1] = h->rhs_gen;
  h->ring_sumsq[0] = h->ring_sumsq[1] = -1.0;
  HIPC(&h->err, hipGetLastError());
  return MG_OK;
}

int mg_update_rhs_device(mg_handle* h, const void* rhs_dev, int ld, int dtype) {
  if (!h || !rhs_dev || !valid_dtype(dtype) || ld < h->lv[0].ny) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_update_rhs_device: bad argument");
  if (!h->have_rhs) return fail(&h->err, MG_ERR_STATE, "mg_update_rhs_device before mg_set_rhs / mg_set_rhs_device");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  Level& v = h->lv[0];
  for (int dt = 0; dt < 2; ++dt)
    if (v.rhs[dt]) d_convert(dtype, dt, rhs_dev, v.rhs[dt], v.nx, v.ny, ld, v.ld[dt], h->stream);
  h->norm_partials = 0;
  // a new right-hand side as far as cached norms go, the same one as far as the coarse rhs rings and their sums go
  const unsigned old_gen = h->rhs_gen++;
  for (int p = 0; p < 2; ++p)
    if (h->rings_gen[p] == old_gen) h->rings_gen[p] = h->rhs_gen;
  HIPC(&h->err, hipGetLastError());
  return MG_OK;
}

int mg_zero_solution_device(mg_handle* h) {
  if (!h) return fail(nullptr, MG_ERR_INVALID_VALUE, "NULL handle");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  h->norm_partials = 0;
  h->iterate_zero = true;
  Level& v = h->lv[0];
  const int dt = h->iterate_dtype();
  HIPC(&h->err, hipMemsetAsync(v.u[dt], 0, (size_t)v.nx * v.ld[dt] * esize(dt), h->stream));
  if (v.t[dt]) HIPC(&h->err, hipMemsetAsync(v.t[dt], 0, (size_t)v.nx * v.ld[dt] * esize(dt), h->stream));
  return MG_OK;
}

int mg_get_solution_device(mg_handle* h, void* u_dev, int ld, int dtype) {
  if (!h || !u_dev || !valid_dtype(dtype) || ld < h->lv[0].ny) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_get_solution_device: bad argument");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  Level& v = h->lv[0];
  const int dt = h->iterate_dtype();
  d_convert(dt, dtype, v.u[dt], u_dev, v.nx, v.ny, v.ld[dt], ld, h->stream);
  HIPC(&h->err, hipGetLastError());
  return MG_OK;
}

int mg_synchronize(mg_handle* h) {
  if (!h) return fail(nullptr, MG_ERR_INVALID_VALUE, "NULL handle");
  HIPC(&h->err, hipStreamSynchronize(h->stream));
  return MG_OK;
}

int mg_set_rhs(mg_handle* h, const void* rhs, int host_dtype) {
  if (!h || !rhs || !valid_dtype(host_dtype)) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_set_rhs: bad argument");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  return set_rhs_impl(h, rhs, host_dtype);
}


int mg_set_coefficient(mg_handle* h, const void* a_host, int host_dtype) {
  if (!h || !valid_dtype(host_dtype)) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_set_coefficient: bad argument");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  if (a_host && h->cfg.precision == MG_PREC_DEFECT)
    return fail(&h->err, MG_ERR_INVALID_VALUE, "mg_set_coefficient: defect correction (MG_PREC_DEFECT) runs the constant-coefficient operator");
  h->norm_partials = 0;
  if (!a_host) {                                           // back to the constant-coefficient operator
    const bool was = h->varcoef;
    h->varcoef = false;
    return was ? plan_tail(h) : MG_OK;
  }
  // The coefficient lives in every precision a level may compute in.  Level l takes every 2^l-th vertex value of the
  // caller's array (injection = re-discretisation), cast ONCE from the caller's dtype to the level's: no level ever
  // sees a value that went through a narrower precision on the way down.
  Level& v0 = h->lv[0];
  const int lds = pitch_elems(host_dtype, v0.ny);
  HIPC(&h->err, hipMemcpy2DAsync(h->staging, (size_t)lds * esize(host_dtype), a_host, (size_t)v0.ny * esize(host_dtype),
                                 (size_t)v0.ny * esize(host_dtype), v0.nx, hipMemcpyHostToDevice, h->stream));
  for (int l = 0; l < h->L(); ++l) {
    Level& v = h->lv[l];
    for (int dt = 0; dt < 2; ++dt) {
      if (!v.u[dt]) continue;
      if (!v.a[dt]) { const int rc = alloc_zero(&h->err, &v.a[dt], (size_t)v.nx * v.ld[dt] * esize(dt), h->stream); if (rc != MG_OK) return rc; }
      if (!v.rd[dt]) { const int rc = alloc_zero(&h->err, &v.rd[dt], (size_t)v.nx * v.ld[dt] * esize(dt), h->stream); if (rc != MG_OK) return rc; }
      d_inject(host_dtype, dt, h->staging, v.a[dt], lds, v.nx, v.ny, v.ld[dt], 1 << l, h->stream);
    }
  }
  const bool was = h->varcoef;
  h->varcoef = true;
  h->rd_sigma = -1.0;                                      // new coefficient: new reciprocal diagonals
  refresh_rdiag(h);
  HIPC(&h->err, hipStreamSynchronize(h->stream));
  h->tail_minv_sigma = -1.0;                               // a direct coarsest solve needs the inverse of the NEW operator
  return was ? MG_OK : plan_tail(h);                       // the LDS tail carries one more array per level
}

#if MG_EXP_TAIL_TRACE
// timing experiment: s_memtime stamps of the last coarse_tail_kernel launch (entry, prologue, after every op, exit)
int mg_exp_tail_trace(long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(mg::g_tail_trace), sizeof(long long) * 64) == hipSuccess ? MG_OK : MG_ERR_HIP;
}
#endif

int mg_set_shift(mg_handle* h, double sigma) {
  if (!h || !(sigma >= 0.0) || !std::isfinite(sigma))
    return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_set_shift: sigma must be finite and >= 0");
  h->sigma = sigma;
  h->norm_partials = 0;     // a cached sum r^2 belongs to the previous operator
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  refresh_rdiag(h);         // variable coefficients: the reciprocal diagonals carry the shift
  return MG_OK;
}

int mg_set_solution(mg_handle* h, const void* u0, int host_dtype) {
  if (!h || !valid_dtype(host_dtype)) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_set_solution: bad argument");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  return set_u_impl(h, u0, host_dtype);
}

int mg_get_solution(mg_handle* h, void* u_out, int host_dtype) {
  if (!h || !u_out || !valid_dtype(host_dtype)) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_get_solution: bad argument");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  Level& v = h->lv[0];
  const int dt = h->iterate_dtype();
  return download(&h->err, u_out, host_dtype, v.u[dt], dt, v.ld[dt], v.nx, v.ny, h->staging, h->stream);
}

int mg_fmg(mg_handle* h, int cycles_per_level) {
  if (!h || cycles_per_level < 0) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_fmg: bad argument");
  if (!h->have_rhs) return fail(&h->err, MG_ERR_STATE, "mg_fmg before mg_set_rhs");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  const int rc = h->cfg.precision == MG_PREC_DEFECT ? defect_fmg(h, cycles_per_level) : fmg_init(h, cycles_per_level);
  if (rc != MG_OK) return fail(&h->err, rc, h->varcoef && h->cfg.precision == MG_PREC_DEFECT
                                                 ? "mg_fmg: defect correction runs the constant-coefficient operator"
                                                 : "mg_fmg: unsupported precision combination");
  return MG_OK;
}

int mg_cycle(mg_handle* h, int ncycles) {
  if (!h || ncycles < 0) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_cycle: bad argument");
  if (!h->have_rhs) return fail(&h->err, MG_ERR_STATE, "mg_cycle before mg_set_rhs");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  for (int k = 0; k < ncycles; ++k) {
    const int rc = run_cycle(h);
    if (rc != MG_OK) return fail(&h->err, rc, "cycle: unsupported precision combination");
  }
  HIPC(&h->err, hipGetLastError());
  return MG_OK;
}

int mg_residual_norm(mg_handle* h, double* out) {
  if (!h || !out) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_residual_norm: bad argument");
  if (!h->have_rhs) return fail(&h->err, MG_ERR_STATE, "mg_residual_norm before mg_set_rhs");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  return fine_norm(h, out);
}

int mg_set_working_precision(mg_handle* h, int dtype) {
  if (!h || !valid_dtype(dtype)) return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "bad argument");
  if (h->cfg.precision != MG_PREC_ADAPTIVE) return fail(&h->err, MG_ERR_STATE, "working precision is fixed unless precision = MG_PREC_ADAPTIVE");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  return switch_phase(h, dtype);
}

static int iterate_impl(mg_handle* h, double tol, int max_iter, double* hist, int hist_cap, int* n_iter,
                        int* converged, int32_t* prec_hist, mg_stats* st) {
  // reset the adaptive state (PrecisionManager starts every solve from default_precision = double)
  if (h->cfg.precision == MG_PREC_ADAPTIVE) {
    const int rc0 = switch_phase(h, MG_F64);
    if (rc0 != MG_OK) return rc0;
    h->promoted = false;
    h->adapt_hist.clear();
  }
  h->fp32_floor = 0.0;
  h->switch_reason = 0;
  h->span_ring[0] = h->span_ring[1] = false;
  for (auto& l : h->lv) l.timings[0] = l.timings[1] = l.timings[2] = 0;
  const double t0 = now_s();
  double rn = 0;
  int rc = fine_norm(h, &rn);
  if (rc != MG_OK) return rc;
  const bool zero_start = h->iterate_zero;    // the first cycle may start from the zero iterate without reading it
  h->iterate_zero = false;                    // cycles follow
  st->initial_residual = rn;
  int it = 0, conv = 0, switches = 0;
  if (h->cfg.precision == MG_PREC_DEFECT) {
    // outer loop of defect correction: fine_norm above left the defect of the initial iterate in the fp32 rhs; each step
    // runs one fp32 cycle from zero on it, then ONE pass that updates the fp64 iterate, forms the next defect and its norm
    if (h->L() < 2) return fail(&h->err, MG_ERR_INVALID_VALUE, "defect correction needs at least two levels");
    inject_rings(h, h->phase);                      // zero rings of every coarse rhs (the error vanishes on the boundary)
    for (it = 1; it <= max_iter; ++it) {
      if ((rc = defect_cycle(h)) != MG_OK) return fail(&h->err, rc, "cycle: unsupported precision combination");
      const int n = launch_defect(h, true);
      double ss = 0;
      if ((rc = reduce_to_host(h, n, &ss)) != MG_OK) return rc;
      rn = std::sqrt(h->lv[0].hx * h->lv[0].hy * ss);
      if (it <= hist_cap) hist[it - 1] = rn;
      if (prec_hist && it <= hist_cap) prec_hist[it - 1] = 3;
      if (rn < tol) { conv = 1; break; }
    }
    if (it > max_iter) it = max_iter;
    HIPC(&h->err, hipMemcpyAsync(h->h_int, h->d_int, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPC(&h->err, hipStreamSynchronize(h->stream));
    st->last_coarse_sweeps = *h->h_int;
    st->solve_seconds = now_s() - t0;
    st->precision_switches = 0;
    if (n_iter) *n_iter = it;
    if (converged) *converged = conv;
    return MG_OK;
  }
  // Speculative launching: while the norm of cycle `it` travels to the host, the FRONT part of cycle it+1 (level-0
  // down leg and everything below it) is already queued -- it never touches the buffer holding the iterate of
  // cycle `it`.  If that norm ends the solve or changes the working precision, the front part is simply dropped
  // (one pointer swap is undone); results are identical to the one-cycle-at-a-time loop.
  const bool can_spec = h->cfg.speculate != 0 && h->fused() && h->L() > 1 && h->mbox_dev && h->cfg.pre <= 2 &&
                        h->cfg.post <= 2 && !h->cfg.profile && h->ring_sumsq[0] >= 0 && h->ring_sumsq[1] >= 0;
  bool spec = false;       // the front part of the coming cycle is already queued
  double prev_rn = 0;      // the norm before `rn` (speculation heuristics)
  const bool zero_first = zero_start && can_spec && h->cfg.pre >= 1;
  auto undo_front = [&]() {
    Level& v0 = h->lv[0];
    const int d0 = h->level_dtype(0);
    std::swap(v0.u[d0], v0.t[d0]);
    spec = false;
  };
  for (it = 1; it <= max_iter; ++it) {
    const int before = h->phase;
    if (spec && h->cfg.precision == MG_PREC_ADAPTIVE) {
      // would the policy switch?  After the speculative swap lv[0].u is the WRONG buffer to convert: drop the queued
      // front part first, then switch from the untouched iterate
      bool promote = false;
      if (adapt_target(h, rn, &promote) != h->phase) undo_front();
      if ((rc = adapt(h, rn)) != MG_OK) return rc;
    } else {
      // a solve from the zero guess: the first cycle's level-0 down leg runs from the zero iterate without reading it
      // (the kernels' ZERO_INIT form, as on every coarse level), so a precision switch before it has nothing to convert
      if ((rc = adapt(h, rn, zero_first && it == 1)) != MG_OK) return rc;               // solvers/multigrid.py:224-227
    }
    if (h->phase != before) {
      ++switches;
      if (h->cfg.adaptive_reference_rule == 0) h->adapt_hist.clear();
    }
    if (can_spec) {
      h->norm_partials = 0;
      if (!spec && (rc = cycle_fused(h, 0, zero_first && it == 1, kPartFront)) != MG_OK) return fail(&h->err, rc, "cycle: unsupported precision combination");
      spec = false;
      // A queued front part is wasted work when the norm in flight switches the working precision (its level-0 leg runs
      // in the old one): extrapolate the norm in flight from the last two of the fp32 phase and do not speculate across a
      // switch the policy would take on it (threshold reached, or the stagnation window filling up with a flat history).
      bool switch_likely = false;
      if (h->cfg.precision == MG_PREC_ADAPTIVE && h->phase == MG_F32 && !h->promoted && !h->cfg.adaptive_reference_rule &&
          h->adapt_hist.size() >= 2) {
        const double prev = h->adapt_hist[h->adapt_hist.size() - 2];
        const double rho = prev > 0 ? std::min(1.0, rn / prev) : 1.0;
        std::vector<double> guess(h->adapt_hist);
        guess.push_back(rn * rho);
        switch_likely = rn * rho < h->cfg.switch_threshold * 10 || stagnating(guess);
      }
      // ... nor across the end of the solve: the norm in flight, extrapolated the same way, meets the tolerance
      if (it >= 2 && prev_rn > 0 && rn * std::min(1.0, rn / prev_rn) < tol) switch_likely = true;
      // ... nor before the fp32 residual floor of this solve is known (floor_due): it is evaluated from the iterate this
      // cycle leaves, right after its norm, and usually ends the fp32 phase there
      if (h->cfg.precision == MG_PREC_ADAPTIVE && h->phase == MG_F32 && !h->promoted && !h->cfg.adaptive_reference_rule &&
          h->fp32_floor == 0.0) switch_likely = true;
      const bool go = it < max_iter && !switch_likely;                 // queue the front part of cycle it + 1 behind this cycle
      unsigned long long seq;
      if (go && span_ok(h)) {
        // up leg of this cycle and down leg of the next in one level-0 launch (cycle_span).  The iterate of THIS cycle is
        // stored unless nothing can end the solve or change the precision on its norm: no tolerance to meet (tol <= 0, a
        // fixed number of cycles) and no adaptive switch pending
        const bool may_switch = h->cfg.precision == MG_PREC_ADAPTIVE && (!h->promoted || h->cfg.adaptive_reference_rule);
        const bool keep_mid = tol > 0 || may_switch;
        if ((rc = cycle_span(h, keep_mid)) != MG_OK) return fail(&h->err, rc, "cycle: unsupported precision combination");
        seq = reduce_post(h, h->norm_partials);
        if ((rc = cycle_below_fine(h)) != MG_OK) return fail(&h->err, rc, "cycle: unsupported precision combination");
        spec = true;
      } else {
        if ((rc = cycle_fused(h, 0, false, kPartBack)) != MG_OK) return fail(&h->err, rc, "cycle: unsupported precision combination");
        seq = reduce_post(h, h->norm_partials);
        if (go) {
          if ((rc = cycle_fused(h, 0, false, kPartFront)) != MG_OK) return fail(&h->err, rc, "cycle: unsupported precision combination");
          spec = true;
        }
      }
      HIPC(&h->err, hipGetLastError());
      double ss = 0;
      if ((rc = reduce_wait(h, seq, &ss)) != MG_OK) return rc;
      const int d0 = h->level_dtype(0);
      prev_rn = rn;
      rn = std::sqrt(h->lv[0].hx * h->lv[0].hy * (ss + h->ring_sumsq[d0]));
      if (spec) h->norm_partials = 0;        // `partials` still describes cycle `it`, but lv[0].u is ahead of it
    } else {
      if ((rc = run_cycle(h)) != MG_OK) return fail(&h->err, rc, "cycle: unsupported precision combination");
      if ((rc = fine_norm(h, &rn)) != MG_OK) return rc;         // multigrid.py:233
    }
    h->adapt_hist.push_back(rn);
    if (h->cfg.precision == MG_PREC_ADAPTIVE && h->phase == MG_F32 && !h->promoted && !h->cfg.adaptive_reference_rule &&
        h->fp32_floor == 0.0 && !spec) {
      // once per solve, after the first fp32 cycle (no front part of the next cycle is queued yet: see `floor_due` above):
      // ||u||_h of the fp32 iterate -> the residual this precision can reach (at_fp32_floor)
      Level& v0 = h->lv[0];
      const int np = d_sumsq(MG_F32, v0.u[MG_F32], h->partials, v0.ld[MG_F32], 0, v0.nx, 0, v0.ny, h->stream);
      double su = 0;
      if ((rc = reduce_to_host(h, np, &su)) != MG_OK) return rc;
      h->norm_partials = 0;                     // `partials` no longer holds this cycle's sum of r^2
      const Coef c0 = coefs(v0.hx, v0.hy, h->sigma);
      h->fp32_floor = 5.9604644775390625e-8 * c0.diag * std::sqrt(v0.hx * v0.hy * su);      // eps32 = 2^-24
    }
    if (it <= hist_cap) hist[it - 1] = rn;
    if (prec_hist && it <= hist_cap)
      prec_hist[it - 1] = (h->cfg.precision == MG_PREC_MIXED_LEVELS) ? 2 : h->level_dtype(0);
    if (rn < tol) { conv = 1; break; }                         // solvers/base.py:134 (absolute)
  }
  if (spec) undo_front();                                      // a queued front part is dropped: lv[0].u is the iterate again
  if (it > max_iter) it = max_iter;
  HIPC(&h->err, hipMemcpyAsync(h->h_int, h->d_int, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPC(&h->err, hipStreamSynchronize(h->stream));
  st->last_coarse_sweeps = *h->h_int;
  st->solve_seconds = now_s() - t0;
  st->precision_switches = switches;
  st->switch_reason = h->switch_reason;
  st->fp32_floor = h->fp32_floor;
  if (n_iter) *n_iter = it;
  if (converged) *converged = conv;
  return MG_OK;
}

int mg_iterate(mg_handle* h, double tol, int max_iter, double* hist, int hist_cap, int* n_iter, int* converged,
               int32_t* prec_hist, mg_stats* stats) {
  if (!h) return fail(nullptr, MG_ERR_STATE, "Multigrid not properly setup or grid mismatch");
  if (max_iter < 1 || (hist_cap > 0 && !hist)) return fail(&h->err, MG_ERR_INVALID_VALUE, "mg_iterate: bad argument");
  if (!h->have_rhs) return fail(&h->err, MG_ERR_STATE, "mg_iterate before mg_set_rhs");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  mg_stats st;
  std::memset(&st, 0, sizeof(st));
  const int rc = iterate_impl(h, tol, max_iter, hist, hist_cap, n_iter, converged, prec_hist, &st);
  if (stats) *stats = st;
  return rc;
}

int mg_solve(mg_handle* h, const void* rhs, const void* u0, void* u_out, int host_dtype, double tol, int max_iter,
             double* hist, int hist_cap, int* n_iter, int* converged, int32_t* prec_hist, mg_stats* stats) {
  if (!h) return fail(nullptr, MG_ERR_STATE, "Multigrid not properly setup or grid mismatch");
  if (!rhs || !u_out || !valid_dtype(host_dtype) || max_iter < 1 || (hist_cap > 0 && !hist))
    return fail(&h->err, MG_ERR_INVALID_VALUE, "mg_solve: bad argument");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  mg_stats st;
  std::memset(&st, 0, sizeof(st));
  if (h->cfg.precision == MG_PREC_ADAPTIVE) h->phase = MG_F64;   // upload into the fp64 iterate
  double t0 = now_s();
  int rc = set_rhs_impl(h, rhs, host_dtype);
  if (rc != MG_OK) return rc;
  rc = set_u_impl(h, u0, host_dtype);
  if (rc != MG_OK) return rc;
  st.h2d_seconds = now_s() - t0;
  if (h->cfg.fmg_cycles > 0 && !u0) {                        // gpu/gpu_solver.py:583: FMG only without an initial guess
    rc = h->cfg.precision == MG_PREC_DEFECT ? defect_fmg(h, h->cfg.fmg_cycles) : fmg_init(h, h->cfg.fmg_cycles);
    if (rc != MG_OK) return fail(&h->err, rc, "fmg: unsupported precision combination");
  }
  rc = iterate_impl(h, tol, max_iter, hist, hist_cap, n_iter, converged, prec_hist, &st);
  if (rc != MG_OK) return rc;
  t0 = now_s();
  Level& v = h->lv[0];
  const int dt = h->iterate_dtype();
  rc = download(&h->err, u_out, host_dtype, v.u[dt], dt, v.ld[dt], v.nx, v.ny, h->staging, h->stream);
  if (rc != MG_OK) return rc;
  st.d2h_seconds = now_s() - t0;
  if (stats) *stats = st;
  return MG_OK;
}

int mg_time_op(mg_handle* h, int op, int level, int dtype, int reps, double* avg_ms) {
  if (!h || !avg_ms || reps < 1 || level < 0 || level >= h->L() || !valid_dtype(dtype))
    return fail(h ? &h->err : nullptr, MG_ERR_INVALID_VALUE, "mg_time_op: bad argument");
  HIPC(&h->err, hipSetDevice(h->cfg.device));
  Level& v = h->lv[level];
  const int dt = dtype;
  if (op != 6 && (!v.u[dt] || !v.rhs[dt])) return fail(&h->err, MG_ERR_STATE, "mg_time_op: level has no arrays of that dtype");
  if ((op == 0 || op == 10) && h->cfg.smoother != MG_JACOBI) return fail(&h->err, MG_ERR_STATE, "mg_time_op: jacobi needs a Jacobi-configured handle");
  if ((op == 0 || (op >= 7 && op <= 9)) && !v.t[dt]) return fail(&h->err, MG_ERR_STATE, "mg_time_op: no ping-pong buffer on this level");
  if ((op == 2 || op == 4 || op == 5 || op == 7 || op == 8) && (level >= h->L() - 1 || !v.r[dt])) return fail(&h->err, MG_ERR_STATE, "mg_time_op: no coarser level");
  static const int exp_nsweep = exp_env("MG_EXP_NSWEEP", 2);
  h->norm_partials = 0;
  if (level == 0 && (op == 0 || op == 1 || (op >= 5 && op <= 9) || op == 12 || op == 13)) h->iterate_zero = false;    // these rewrite the fine iterate
  // op 10: the single-sweep Jacobi kernel rotating over independent {u, rhs, out} sets whose total exceeds three
  // times the 256 MiB Infinity Cache, so that no launch finds its operands on die: the HBM-proper smoother figure
  std::vector<void*> hbm_sets;
  struct FreeSets { std::vector<void*>& v; ~FreeSets() { for (void* p : v) (void)hipFree(p); } } free_sets{hbm_sets};
  int nsets = 0, set_idx = 0;
  if (op == 10 || op == 11) {
    const size_t bytes = (size_t)v.nx * v.ld[dt] * esize(dt);
    nsets = std::max<int>(3, (int)((768ull << 20) / (3 * bytes)) + 1);
    for (int k = 0; k < 3 * nsets; ++k) {
      void* p = nullptr;
      HIPC(&h->err, hipMalloc(&p, bytes));
      hbm_sets.push_back(p);
      HIPC(&h->err, hipMemcpyAsync(p, (k % 3 == 1) ? v.rhs[dt] : v.u[dt], bytes, hipMemcpyDeviceToDevice, h->stream));
    }
  }
  hipEvent_t e0, e1;
  HIPC(&h->err, hipEventCreate(&e0));
  HIPC(&h->err, hipEventCreate(&e1));
  auto run = [&](int n) -> int {
    for (int k = 0; k < n; ++k) {
      switch (op) {
        case 10: { void** b = hbm_sets.data() + 3 * (set_idx++ % nsets);
                   d_jacobi(dt, b[0], b[1], b[2], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.omega, h->stream, level == 0); } break;
        case 11: { void** b = hbm_sets.data() + 3 * (set_idx++ % nsets);       // the bare stream: same traffic, no stencil
                   const int N = (int)(16 / esize(dt)), nyv = std::min(v.ld[dt], (v.ny + N - 1) / N * N);
                   const int tiles_j = (nyv / N + 63) / 64, tiles_i = (v.nx + 15) / 16;
                   if (dt == MG_F32) hipLaunchKernelGGL(mg::stream_triad_kernel<float>, dim3(tiles_i * tiles_j), dim3(256), 0, h->stream, (const float*)b[0], (const float*)b[1], (float*)b[2], v.nx, nyv, v.ld[dt], tiles_j);
                   else hipLaunchKernelGGL(mg::stream_triad_kernel<double>, dim3(tiles_i * tiles_j), dim3(256), 0, h->stream, (const double*)b[0], (const double*)b[1], (double*)b[2], v.nx, nyv, v.ld[dt], tiles_j); } break;
        case 0: d_jacobi(dt, v.u[dt], v.rhs[dt], v.t[dt], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.omega, h->stream, level == 0);
                std::swap(v.u[dt], v.t[dt]); break;
        case 1: for (int c = 0; c < 2; ++c) d_rbgs_colour(dt, v.u[dt], v.rhs[dt], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.omega, c, h->cfg.colour_offset, h->stream, level == 0); break;
        case 2: d_residual(dt, v.u[dt], v.rhs[dt], v.r[dt], v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.coeff, h->stream, level == 0); break;
        case 3: { const int n2 = d_residual_norm(dt, v.u[dt], v.rhs[dt], h->partials, v.nx, v.ny, v.ld[dt], v.hx, v.hy, h->cfg.coeff, h->stream, level == 0);
                  launch_reduce(h->partials, n2, h->d_scalar, h->stream); } break;
        case 4: { Level& c = h->lv[level + 1]; const int dc = c.rhs[dt] ? dt : 1 - dt;
                  d_restrict(dt, dc, v.r[dt], c.rhs[dc], v.nx, v.ny, v.ld[dt], c.ld[dc], h->stream); } break;
        case 5: { Level& c = h->lv[level + 1]; const int dc = c.u[dt] ? dt : 1 - dt;
                  if (d_prolong<true>(dc, dt, h->grid_dtype, c.u[dc], v.u[dt], v.nx, v.ny, v.ld[dt], c.ld[dc], h->stream) != MG_OK) return MG_ERR_INVALID_VALUE; } break;
        case 6: { const int rc = run_cycle(h); if (rc != MG_OK) return rc; } break;
        case 7: { Level& c = h->lv[level + 1]; const int dc = c.rhs[dt] ? dt : 1 - dt;          // down leg
                  LegGeom g{v.nx, v.ny, v.ld[dt], c.nx, c.ny, c.ld[dc], v.hx, v.hy, h->cfg.omega, h->cfg.coeff, exp_nsweep, h->cfg.colour_offset, level == 0}; g.sigma = h->sigma; g.rb = rb_mode(h); if (h->varcoef) { g.acoef = v.a[dt]; g.rdiag = v.rd[dt]; }
                  d_down(h->cfg.smoother, dt, dc, v.u[dt], v.rhs[dt], v.t[dt], c.rhs[dc], g, false, h->stream);
                  std::swap(v.u[dt], v.t[dt]); } break;
        case 8: { Level& c = h->lv[level + 1]; const int dc = c.u[dt] ? dt : 1 - dt;            // up leg (+ norm on level 0)
                  LegGeom g{v.nx, v.ny, v.ld[dt], c.nx, c.ny, c.ld[dc], v.hx, v.hy, h->cfg.omega, h->cfg.coeff, exp_nsweep, h->cfg.colour_offset, level == 0}; g.sigma = h->sigma; g.rb = rb_mode(h); if (h->varcoef) { g.acoef = v.a[dt]; g.rdiag = v.rd[dt]; }
                  if (d_up(h->cfg.smoother, dt, dc, h->grid_dtype, v.u[dt], v.rhs[dt], v.t[dt], c.u[dc], h->partials, g, level == 0, h->stream) < 0) return MG_ERR_INVALID_VALUE;
                  std::swap(v.u[dt], v.t[dt]); } break;
        case 9: { LegGeom g{v.nx, v.ny, v.ld[dt], 0, 0, 0, v.hx, v.hy, h->cfg.omega, h->cfg.coeff, exp_nsweep, h->cfg.colour_offset, level == 0}; g.sigma = h->sigma; g.rb = rb_mode(h); if (h->varcoef) { g.acoef = v.a[dt]; g.rdiag = v.rd[dt]; }
                  d_sweeps(h->cfg.smoother, dt, v.u[dt], v.rhs[dt], v.t[dt], g, h->stream);
                  std::swap(v.u[dt], v.t[dt]); } break;
        case 12: case 13: {                                                                      // spanning leg, with / without the store of the iterate in between
                  const int phase0 = h->phase;
                  if (h->cfg.precision == MG_PREC_ADAPTIVE) h->phase = dt;                       // time the leg of either working precision
                  const bool ok = level == 0 && span_ok(h) && h->level_dtype(0) == dt;
                  const int rc2 = ok ? cycle_span(h, op == 12) : MG_ERR_INVALID_VALUE;
                  h->phase = phase0;
                  if (rc2 != MG_OK) return rc2; } break;
        default: return MG_ERR_INVALID_VALUE;
      }
    }
    return MG_OK;
  };
  int rc = run(1);   // warm-up
  if (rc != MG_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return fail(&h->err, rc, "mg_time_op: unsupported op"); }
  HIPC(&h->err, hipEventRecord(e0, h->stream));
  rc = run(reps);
  HIPC(&h->err, hipEventRecord(e1, h->stream));
  HIPC(&h->err, hipEventSynchronize(e1));
  float ms = 0;
  HIPC(&h->err, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *avg_ms = (double)ms / reps;
  return rc;
}

// ---------------------------------------------------------------- stateless, device arrays ----
#define CHECK_DEV(cond, msg) do { if (!(cond)) return fail(nullptr, MG_ERR_INVALID_VALUE, msg); } while (0)

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static bool ld_ok(int dt, int ny, int ld) { return ld >= ny && ((size_t)ld * esize(dt)) % 16 == 0; }

int mg_dev_jacobi(int dtype, int nx, int ny, int ld, double hx, double hy, double omega, const void* u, const void* rhs,
                  void* out, void* stream) {
  CHECK_DEV(valid_dtype(dtype) && nx >= 3 && ny >= 3 && ld_ok(dtype, ny, ld), "mg_dev_jacobi: bad shape / pitch");
  CHECK_DEV(u && rhs && out && u != out && aligned16(u) && aligned16(rhs) && aligned16(out), "mg_dev_jacobi: bad pointer");
  d_jacobi(dtype, u, rhs, out, nx, ny, ld, hx, hy, omega, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_rbgs_colour(int dtype, int nx, int ny, int ld, double hx, double hy, double omega, int colour, int colour_offset,
                       void* u, const void* rhs, void* stream) {
  CHECK_DEV(valid_dtype(dtype) && nx >= 3 && ny >= 3 && ld_ok(dtype, ny, ld) && (colour == 0 || colour == 1), "mg_dev_rbgs_colour: bad argument");
  CHECK_DEV(u && rhs && aligned16(u) && aligned16(rhs), "mg_dev_rbgs_colour: bad pointer");
  d_rbgs_colour(dtype, u, rhs, nx, ny, ld, hx, hy, omega, colour, colour_offset, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_residual(int dtype, int nx, int ny, int ld, double hx, double hy, double coeff, const void* u, const void* f,
                    void* r, void* stream) {
  CHECK_DEV(valid_dtype(dtype) && nx >= 3 && ny >= 3 && ld_ok(dtype, ny, ld), "mg_dev_residual: bad shape / pitch");
  CHECK_DEV(u && f && r && aligned16(u) && aligned16(f) && aligned16(r), "mg_dev_residual: bad pointer");
  d_residual(dtype, u, f, r, nx, ny, ld, hx, hy, coeff, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_residual_f32in_f64out(int nx, int ny, int ld_in, int ld_out, double hx, double hy, double coeff, const float* u,
                                 const float* f, double* r, void* stream) {
  CHECK_DEV(nx >= 3 && ny >= 3 && ld_in >= ny && ld_out >= ny, "mg_dev_residual_f32in_f64out: bad shape / pitch");
  CHECK_DEV(u && f && r, "mg_dev_residual_f32in_f64out: bad pointer");
  const Coef c = coefs(hx, hy);
  const long long pairs = (long long)nx * ((ny + 1) / 2);
  hipLaunchKernelGGL((mg::residual_xprec_kernel<float, float, double, false, false, false>), dim3(grid_for(pairs)), dim3(mg::kBlock), 0,
                     (hipStream_t)stream, u, (const float*)nullptr, f, r, (double*)nullptr, (double*)nullptr, nx, ny, ld_in, ld_in, ld_in,
                     ld_out, c.ihx2, c.ihy2, c.diag, coeff);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_scratch_bytes(int nx, int ny, int64_t* bytes) {
  CHECK_DEV(bytes && nx >= 1 && ny >= 1, "mg_dev_scratch_bytes: bad argument");
  *bytes = (int64_t)sizeof(double) * (int64_t)max_partials(nx, ny);
  return MG_OK;
}

int mg_dev_sumsq(int dtype, int ld, int i_lo, int i_hi, int j_lo, int j_hi, const void* field, void* scratch,
                 double* sumsq_dev, void* stream) {
  CHECK_DEV(valid_dtype(dtype) && i_lo >= 0 && i_hi >= i_lo && j_lo >= 0 && j_hi >= j_lo && ld_ok(dtype, j_hi, ld), "mg_dev_sumsq: bad window / pitch");
  CHECK_DEV(field && scratch && sumsq_dev && aligned16(field), "mg_dev_sumsq: bad pointer");
  const int n = d_sumsq(dtype, field, (double*)scratch, ld, i_lo, i_hi, j_lo, j_hi, (hipStream_t)stream);
  launch_reduce((double*)scratch, n, sumsq_dev, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_restrict_fw(int in_dtype, int out_dtype, int nxf, int nyf, int ldf, int nxc, int nyc, int ldc, int sides,
                       const void* fine, void* coarse, void* stream) {
  CHECK_DEV(valid_dtype(in_dtype) && valid_dtype(out_dtype) && nxf >= 3 && nyf >= 3 && nxc >= 2 && nyc >= 2 && sides >= 0 && sides <= 15, "mg_dev_restrict_fw: bad argument");
  // interior coarse cells read fine rows/cols 2c-1..2c+1; a physical far edge is injected from fine 2(nc-1)
  CHECK_DEV(2 * (nxc - 2) + 1 <= nxf - 1 && 2 * (nyc - 2) + 1 <= nyf - 1, "Cannot restrict: coarse grid too large for the fine grid");
  CHECK_DEV((!(sides & 2) || 2 * (nxc - 1) <= nxf - 1) && (!(sides & 8) || 2 * (nyc - 1) <= nyf - 1), "Cannot restrict: coarse boundary outside the fine grid");
  CHECK_DEV(ld_ok(in_dtype, nyf, ldf) && ld_ok(out_dtype, nyc, ldc) && fine && coarse && aligned16(coarse), "mg_dev_restrict_fw: bad pitch / pointer");
  d_restrict_sub(in_dtype, out_dtype, fine, coarse, ldf, nxc, nyc, ldc, sides, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_prolong_add(int coarse_dtype, int fine_dtype, int compute_dtype, int nxf, int nyf, int ldf, int nxc, int nyc,
                       int ldc, int sides, const void* coarse, void* fine_u, void* stream) {
  CHECK_DEV(valid_dtype(coarse_dtype) && valid_dtype(fine_dtype) && valid_dtype(compute_dtype) && nxf >= 3 && nyf >= 3 && nxc >= 2 && nyc >= 2 && sides >= 0 && sides <= 15, "mg_dev_prolong_add: bad argument");
  CHECK_DEV(ld_ok(fine_dtype, nyf, ldf) && ldc >= nyc && coarse && fine_u && aligned16(fine_u), "mg_dev_prolong_add: bad pitch / pointer");
  const int rc = d_prolong_sub<true>(coarse_dtype, fine_dtype, compute_dtype, coarse, fine_u, nxf, nyf, ldf, nxc, nyc, ldc, sides, (hipStream_t)stream);
  if (rc != MG_OK) return fail(nullptr, rc, "mg_dev_prolong_add: fp32 interpolation needs fp32 coarse and fine fields");
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

// ---- fused legs on device arrays (sub-domains: wide ghost zones, coarse offsets, norm window) --------------------
int mg_dev_down_leg(int smoother, int dtype, int coarse_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc, int ci_off,
                    int cj_off, double hx, double hy, double omega, double coeff, int nsweep, int zero_init, int colour_offset,
                    const void* u, const void* rhs, void* out, void* rhs_coarse, void* stream, int select, const int* inner_rect) {
  return mg_dev_down_leg_var(smoother, dtype, coarse_dtype, nx, ny, ld, nxc, nyc, ldc, ci_off, cj_off, hx, hy, omega, coeff, nsweep,
                             zero_init, colour_offset, u, rhs, out, rhs_coarse, stream, select, inner_rect, nullptr, nullptr);
}

int mg_dev_down_leg_var(int smoother, int dtype, int coarse_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc, int ci_off,
                        int cj_off, double hx, double hy, double omega, double coeff, int nsweep, int zero_init, int colour_offset,
                        const void* u, const void* rhs, void* out, void* rhs_coarse, void* stream, int select, const int* inner_rect,
                        const void* acoef, const void* rdiag) {
  CHECK_DEV((smoother == MG_JACOBI || smoother == MG_RBGS) && valid_dtype(dtype) && valid_dtype(coarse_dtype), "mg_dev_down_leg: bad smoother / dtype");
  CHECK_DEV((!acoef && !rdiag) || (acoef && rdiag && aligned16(acoef) && aligned16(rdiag)), "mg_dev_down_leg: coefficient and reciprocal diagonal come together, 16-byte aligned");
  CHECK_DEV(nx >= 3 && ny >= 3 && nxc >= 3 && nyc >= 3 && ld_ok(dtype, ny, ld) && ldc >= nyc && nsweep >= 0 && nsweep <= 2, "mg_dev_down_leg: bad shape / pitch / sweep count");
  CHECK_DEV(rhs && out && rhs_coarse && (zero_init || u) && u != out && aligned16(rhs) && aligned16(out) && (!u || aligned16(u)), "mg_dev_down_leg: bad pointer");
  LegGeom g{nx, ny, ld, nxc, nyc, ldc, hx, hy, omega, coeff, nsweep, colour_offset, false};
  g.ci_off = ci_off; g.cj_off = cj_off;
  CHECK_DEV(select >= 0 && select <= 2 && (select == 0 || inner_rect), "mg_dev_down_leg: bad tile selection");
  if (select) { g.select = select; g.in_i_lo = inner_rect[0]; g.in_i_hi = inner_rect[1]; g.in_j_lo = inner_rect[2]; g.in_j_hi = inner_rect[3]; }
  g.acoef = acoef; g.rdiag = rdiag;
  g.rb = 1;                       // register-blocked legs on blocks above ~1100^2 cells (same results, mg_config.fused = 2)
  d_down(smoother, dtype, coarse_dtype, u ? u : rhs, rhs, out, rhs_coarse, g, zero_init != 0, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_up_leg(int smoother, int dtype, int coarse_dtype, int compute_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc,
                  int ci_off, int cj_off, int sides, double hx, double hy, double omega, double coeff, int nsweep, int colour_offset,
                  const void* u, const void* rhs, void* out, const void* e_coarse, int norm, int ni_lo, int ni_hi, int nj_lo,
                  int nj_hi, void* scratch, double* sumsq_dev, void* stream) {
  return mg_dev_up_leg_var(smoother, dtype, coarse_dtype, compute_dtype, nx, ny, ld, nxc, nyc, ldc, ci_off, cj_off, sides, hx, hy, omega,
                           coeff, nsweep, colour_offset, u, rhs, out, e_coarse, norm, ni_lo, ni_hi, nj_lo, nj_hi, scratch, sumsq_dev,
                           stream, nullptr, nullptr);
}

int mg_dev_up_leg_var(int smoother, int dtype, int coarse_dtype, int compute_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc,
                      int ci_off, int cj_off, int sides, double hx, double hy, double omega, double coeff, int nsweep, int colour_offset,
                      const void* u, const void* rhs, void* out, const void* e_coarse, int norm, int ni_lo, int ni_hi, int nj_lo,
                      int nj_hi, void* scratch, double* sumsq_dev, void* stream, const void* acoef, const void* rdiag) {
  CHECK_DEV((smoother == MG_JACOBI || smoother == MG_RBGS) && valid_dtype(dtype) && valid_dtype(coarse_dtype) && valid_dtype(compute_dtype), "mg_dev_up_leg: bad smoother / dtype");
  CHECK_DEV((!acoef && !rdiag) || (acoef && rdiag && aligned16(acoef) && aligned16(rdiag)), "mg_dev_up_leg: coefficient and reciprocal diagonal come together, 16-byte aligned");
  CHECK_DEV(nx >= 3 && ny >= 3 && nxc >= 2 && nyc >= 2 && ld_ok(dtype, ny, ld) && ldc >= nyc && nsweep >= 0 && nsweep <= 2 && sides >= 0 && sides <= 15, "mg_dev_up_leg: bad shape / pitch / sweep count");
  CHECK_DEV(u && rhs && out && e_coarse && u != out && aligned16(u) && aligned16(rhs) && aligned16(out) && (!norm || (scratch && sumsq_dev)), "mg_dev_up_leg: bad pointer");
  LegGeom g{nx, ny, ld, nxc, nyc, ldc, hx, hy, omega, coeff, nsweep, colour_offset, false};
  g.ci_off = ci_off; g.cj_off = cj_off; g.sides = sides;
  if (norm) { g.ni_lo = ni_lo; g.ni_hi = ni_hi; g.nj_lo = nj_lo; g.nj_hi = nj_hi; }
  g.acoef = acoef; g.rdiag = rdiag;
  g.rb = 1;
  const int n = d_up(smoother, dtype, coarse_dtype, compute_dtype, u, rhs, out, e_coarse, (double*)scratch, g, norm != 0, (hipStream_t)stream);
  if (n < 0) return fail(nullptr, MG_ERR_INVALID_VALUE, "mg_dev_up_leg: fp32 interpolation needs fp32 coarse and fine fields");
  if (norm) launch_reduce((double*)scratch, n, sumsq_dev, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_span_leg_ok(int smoother, int dtype, int coarse_dtype, int compute_dtype, int nx, int ny) {
  if (smoother != MG_JACOBI || !valid_dtype(dtype) || coarse_dtype != dtype || !valid_dtype(compute_dtype)) return 0;
  if (dtype == MG_F64 && compute_dtype != MG_F64) return 0;
  LegGeom g{nx, ny, 0, 0, 0, 0, 1.0, 1.0, 0, 0, 0, 0, false};
  g.rb = 1;
  return use_rb(g, mg::kSmJacobi) ? 1 : 0;
}

int mg_dev_span_leg(int smoother, int dtype, int coarse_dtype, int compute_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc,
                    int ci_off, int cj_off, int sides, double hx, double hy, double omega, double coeff, int nsweep_post,
                    int nsweep_pre, int colour_offset, const void* u, const void* rhs, void* out_mid, void* out_next,
                    const void* e_coarse, void* rhs_coarse, int ni_lo, int ni_hi, int nj_lo, int nj_hi, void* scratch,
                    double* sumsq_dev, void* stream) {
  CHECK_DEV(mg_dev_span_leg_ok(smoother, dtype, coarse_dtype, compute_dtype, nx, ny), "mg_dev_span_leg: weighted Jacobi on one dtype and arrays above ~1100^2 cells only");
  CHECK_DEV(nx >= 3 && ny >= 3 && nxc >= 2 && nyc >= 2 && ld_ok(dtype, ny, ld) && ldc >= nyc && nsweep_post >= 1 && nsweep_post <= 2 &&
            nsweep_pre >= 1 && nsweep_pre <= 2 && sides >= 0 && sides <= 15, "mg_dev_span_leg: bad shape / pitch / sweep count");
  CHECK_DEV(u && rhs && out_next && e_coarse && rhs_coarse && scratch && sumsq_dev && u != out_next && u != out_mid && out_mid != out_next &&
            aligned16(u) && aligned16(rhs) && aligned16(out_next) && (!out_mid || aligned16(out_mid)), "mg_dev_span_leg: bad pointer");
  LegGeom g{nx, ny, ld, nxc, nyc, ldc, hx, hy, omega, coeff, nsweep_post, colour_offset, false};
  g.ci_off = ci_off; g.cj_off = cj_off; g.sides = sides;
  g.ni_lo = ni_lo; g.ni_hi = ni_hi; g.nj_lo = nj_lo; g.nj_hi = nj_hi;
  g.rb = 1;
  const int n = d_span(dtype, compute_dtype, u, rhs, out_mid, out_next, e_coarse, rhs_coarse, (double*)scratch, g, nsweep_pre, (hipStream_t)stream);
  if (n < 0) return fail(nullptr, MG_ERR_INVALID_VALUE, "mg_dev_span_leg: unsupported precision combination");
  launch_reduce((double*)scratch, n, sumsq_dev, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_var_rdiag(int dtype, int nx, int ny, int ld, double hx, double hy, double sigma, const void* a, void* rdiag, void* stream) {
  CHECK_DEV(valid_dtype(dtype) && nx >= 3 && ny >= 3 && ld_ok(dtype, ny, ld) && sigma >= 0.0, "mg_dev_var_rdiag: bad shape / pitch / shift");
  CHECK_DEV(a && rdiag && a != rdiag, "mg_dev_var_rdiag: bad pointer");
  d_rdiag(dtype, a, rdiag, nx, ny, ld, hx, hy, sigma, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_inject_ring(int in_dtype, int out_dtype, int nxf, int nyf, int ldf, int nxc, int nyc, int ldc, int sides, int ci_off,
                       int cj_off, const void* fine, void* coarse, void* stream) {
  CHECK_DEV(valid_dtype(in_dtype) && valid_dtype(out_dtype) && nxf >= 3 && nyf >= 3 && nxc >= 2 && nyc >= 2 && ldf >= nyf && ldc >= nyc && fine && coarse && sides >= 0 && sides <= 15, "mg_dev_inject_ring: bad argument");
  d_inject_ring(in_dtype, out_dtype, fine, coarse, nxf, nyf, ldf, nxc, nyc, ldc, (hipStream_t)stream, sides, ci_off, cj_off);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

int mg_dev_convert(int in_dtype, int out_dtype, int nx, int ny, int ldi, int ldo, const void* in, void* out, void* stream) {
  CHECK_DEV(valid_dtype(in_dtype) && valid_dtype(out_dtype) && nx >= 1 && ny >= 1 && ldi >= ny && ldo >= ny && in && out, "mg_dev_convert: bad argument");
  d_convert(in_dtype, out_dtype, in, out, nx, ny, ldi, ldo, (hipStream_t)stream);
  HIPC(nullptr, hipGetLastError());
  return MG_OK;
}

// ---------------------------------------------------------------- stateless, host arrays ------
namespace {
struct Scratch {   // device buffers of one host-pointer call, freed on scope exit
  std::vector<void*> ptrs;
  hipStream_t st = nullptr;
  ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
  int get(void** p, int dt, int nx, int ny) {
    const size_t bytes = (size_t)nx * pitch_elems(dt, ny) * esize(dt);
    const int rc = alloc_zero(nullptr, p, bytes);
    if (rc == MG_OK) ptrs.push_back(*p);
    return rc;
  }
};
int need_device() {
  int n = 0;
  const int rc = mg_device_count(&n);
  if (rc != MG_OK) return rc;
  if (n <= 0) return fail(nullptr, MG_ERR_NO_DEVICE, "no HIP device visible");
  return MG_OK;
}
int up(void* dev, int dt, const void* host, int nx, int ny) {
  const int ld = pitch_elems(dt, ny);
  HIPC(nullptr, hipMemcpy2D(dev, (size_t)ld * esize(dt), host, (size_t)ny * esize(dt), (size_t)ny * esize(dt), nx, hipMemcpyHostToDevice));
  return MG_OK;
}
int down(void* host, int dt, const void* dev, int nx, int ny) {
  const int ld = pitch_elems(dt, ny);
  HIPC(nullptr, hipDeviceSynchronize());
  HIPC(nullptr, hipMemcpy2D(host, (size_t)ny * esize(dt), dev, (size_t)ld * esize(dt), (size_t)ny * esize(dt), nx, hipMemcpyDeviceToHost));
  return MG_OK;
}
#define RC(x) do { const int rc_ = (x); if (rc_ != MG_OK) return rc_; } while (0)
}  // namespace

int mg_op_residual(int dtype, int nx, int ny, double hx, double hy, double coeff, const void* u, const void* f, void* r) {
  CHECK_DEV(valid_dtype(dtype) && u && f && r, "mg_op_residual: bad argument");
  CHECK_DEV(nx >= 3 && ny >= 3, "Cannot apply Laplacian to grid");   // operators/laplacian.py:55-56
  RC(need_device());
  Scratch s; void *du, *df, *dr;
  RC(s.get(&du, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny)); RC(s.get(&dr, dtype, nx, ny));
  RC(up(du, dtype, u, nx, ny)); RC(up(df, dtype, f, nx, ny));
  d_residual(dtype, du, df, dr, nx, ny, pitch_elems(dtype, ny), hx, hy, coeff, nullptr);
  return down(r, dtype, dr, nx, ny);
}

int mg_op_residual_mixed(int nx, int ny, double hx, double hy, double coeff, const float* u, const float* f, double* r) {
  CHECK_DEV(u && f && r, "mg_op_residual_mixed: bad argument");
  CHECK_DEV(nx >= 3 && ny >= 3, "Cannot apply Laplacian to grid");   // operators/laplacian.py:55-56
  RC(need_device());
  Scratch s; void *du, *df, *dr;
  RC(s.get(&du, MG_F32, nx, ny)); RC(s.get(&df, MG_F32, nx, ny)); RC(s.get(&dr, MG_F64, nx, ny));
  RC(up(du, MG_F32, u, nx, ny)); RC(up(df, MG_F32, f, nx, ny));
  RC(mg_dev_residual_f32in_f64out(nx, ny, pitch_elems(MG_F32, ny), pitch_elems(MG_F64, ny), hx, hy, coeff, (const float*)du, (const float*)df,
                                  (double*)dr, nullptr));
  return down(r, MG_F64, dr, nx, ny);
}

int mg_op_apply(int dtype, int nx, int ny, double hx, double hy, double coeff, const void* u, void* au) {
  CHECK_DEV(valid_dtype(dtype) && u && au, "mg_op_apply: bad argument");
  CHECK_DEV(nx >= 3 && ny >= 3, "Cannot apply Laplacian to grid");
  RC(need_device());
  // A u = 0 - (0 - A u): residual of a zero right-hand side under the negated coefficient (exact sign flips)
  Scratch s; void *du, *df, *dr;
  RC(s.get(&du, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny)); RC(s.get(&dr, dtype, nx, ny));
  RC(up(du, dtype, u, nx, ny));
  d_residual(dtype, du, df, dr, nx, ny, pitch_elems(dtype, ny), hx, hy, -coeff, nullptr);
  return down(au, dtype, dr, nx, ny);
}

int mg_op_norm(int dtype, int nx, int ny, double hx, double hy, const void* field, double* out) {
  CHECK_DEV(valid_dtype(dtype) && field && out && nx >= 1 && ny >= 1, "mg_op_norm: bad argument");
  RC(need_device());
  Scratch s; void* dfld; void* part; void* acc;
  RC(s.get(&dfld, dtype, nx, ny));
  RC(alloc_zero(nullptr, &part, sizeof(double) * 2048)); s.ptrs.push_back(part);
  RC(alloc_zero(nullptr, &acc, sizeof(double))); s.ptrs.push_back(acc);
  RC(up(dfld, dtype, field, nx, ny));
  const int n = d_sumsq(dtype, dfld, (double*)part, pitch_elems(dtype, ny), 0, nx, 0, ny, nullptr);
  launch_reduce((double*)part, n, (double*)acc, nullptr);
  double ss = 0;
  HIPC(nullptr, hipMemcpy(&ss, acc, sizeof(double), hipMemcpyDeviceToHost));
  *out = std::sqrt(hx * hy * ss);
  return MG_OK;
}

int mg_op_jacobi(int dtype, int nx, int ny, double hx, double hy, double omega, int nu, const void* u, const void* rhs, void* out) {
  CHECK_DEV(valid_dtype(dtype) && u && rhs && out && nu >= 0 && nx >= 3 && ny >= 3, "mg_op_jacobi: bad argument");
  RC(need_device());
  Scratch s; void *da, *db, *df;
  RC(s.get(&da, dtype, nx, ny)); RC(s.get(&db, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny));
  RC(up(da, dtype, u, nx, ny)); RC(up(db, dtype, u, nx, ny)); RC(up(df, dtype, rhs, nx, ny));
  for (int k = 0; k < nu; ++k) { d_jacobi(dtype, da, df, db, nx, ny, pitch_elems(dtype, ny), hx, hy, omega, nullptr); std::swap(da, db); }
  return down(out, dtype, da, nx, ny);
}

int mg_op_rbgs(int dtype, int nx, int ny, double hx, double hy, double omega, int nu, const void* u, const void* rhs, void* out) {
  CHECK_DEV(valid_dtype(dtype) && u && rhs && out && nu >= 0 && nx >= 3 && ny >= 3, "mg_op_rbgs: bad argument");
  RC(need_device());
  Scratch s; void *da, *df;
  RC(s.get(&da, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny));
  RC(up(da, dtype, u, nx, ny)); RC(up(df, dtype, rhs, nx, ny));
  for (int k = 0; k < nu; ++k)
    for (int c = 0; c < 2; ++c) d_rbgs_colour(dtype, da, df, nx, ny, pitch_elems(dtype, ny), hx, hy, omega, c, 0, nullptr);
  return down(out, dtype, da, nx, ny);
}


int mg_op_helmholtz(int dtype, int op, int nx, int ny, double hx, double hy, double coeff, double sigma, double omega, int nu,
                    const void* u, const void* f, void* out) {
  CHECK_DEV(valid_dtype(dtype) && u && f && out && nu >= 0 && nx >= 3 && ny >= 3 && op >= 0 && op <= 2 && sigma >= 0.0,
            "mg_op_helmholtz: bad argument");
  RC(need_device());
  const int ld = pitch_elems(dtype, ny);
  Scratch s; void *da, *db, *df;
  RC(s.get(&da, dtype, nx, ny)); RC(s.get(&db, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny));
  RC(up(da, dtype, u, nx, ny)); RC(up(db, dtype, u, nx, ny)); RC(up(df, dtype, f, nx, ny));
  if (op == 0) {
    d_residual(dtype, da, df, db, nx, ny, ld, hx, hy, coeff, nullptr, false, sigma);
    return down(out, dtype, db, nx, ny);
  }
  for (int k = 0; k < nu; ++k) {
    if (op == 1) { d_jacobi(dtype, da, df, db, nx, ny, ld, hx, hy, omega, nullptr, false, sigma); std::swap(da, db); }
    else for (int c = 0; c < 2; ++c) d_rbgs_colour(dtype, da, df, nx, ny, ld, hx, hy, omega, c, 0, nullptr, false, sigma);
  }
  return down(out, dtype, da, nx, ny);
}

int mg_op_residual_var(int dtype, int nx, int ny, double hx, double hy, double coeff, const void* a, const void* u, const void* f, void* r) {
  CHECK_DEV(valid_dtype(dtype) && a && u && f && r && nx >= 3 && ny >= 3, "mg_op_residual_var: bad argument");
  RC(need_device());
  Scratch s; void *da, *du, *df, *dr;
  RC(s.get(&da, dtype, nx, ny)); RC(s.get(&du, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny)); RC(s.get(&dr, dtype, nx, ny));
  RC(up(da, dtype, a, nx, ny)); RC(up(du, dtype, u, nx, ny)); RC(up(df, dtype, f, nx, ny));
  d_var<mg::kVarResidual>(dtype, du, da, df, dr, nullptr, nx, ny, pitch_elems(dtype, ny), hx, hy, 1.0, coeff, 0, 0, nullptr);
  return down(r, dtype, dr, nx, ny);
}

int mg_op_jacobi_var(int dtype, int nx, int ny, double hx, double hy, double omega, int nu, const void* a, const void* u, const void* rhs, void* out) {
  CHECK_DEV(valid_dtype(dtype) && a && u && rhs && out && nu >= 0 && nx >= 3 && ny >= 3, "mg_op_jacobi_var: bad argument");
  RC(need_device());
  Scratch s; void *dc, *da, *db, *df;
  RC(s.get(&dc, dtype, nx, ny)); RC(s.get(&da, dtype, nx, ny)); RC(s.get(&db, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny));
  RC(up(dc, dtype, a, nx, ny)); RC(up(da, dtype, u, nx, ny)); RC(up(db, dtype, u, nx, ny)); RC(up(df, dtype, rhs, nx, ny));
  for (int k = 0; k < nu; ++k) {
    d_var<mg::kVarJacobi>(dtype, da, dc, df, db, nullptr, nx, ny, pitch_elems(dtype, ny), hx, hy, omega, -1.0, 0, 0, nullptr);
    std::swap(da, db);
  }
  return down(out, dtype, da, nx, ny);
}

int mg_op_rbgs_var(int dtype, int nx, int ny, double hx, double hy, double omega, int nu, const void* a, const void* u, const void* rhs, void* out) {
  CHECK_DEV(valid_dtype(dtype) && a && u && rhs && out && nu >= 0 && nx >= 3 && ny >= 3, "mg_op_rbgs_var: bad argument");
  RC(need_device());
  Scratch s; void *dc, *da, *df;
  RC(s.get(&dc, dtype, nx, ny)); RC(s.get(&da, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny));
  RC(up(dc, dtype, a, nx, ny)); RC(up(da, dtype, u, nx, ny)); RC(up(df, dtype, rhs, nx, ny));
  for (int k = 0; k < nu; ++k)
    for (int c = 0; c < 2; ++c)
      d_var<mg::kVarRbgs>(dtype, da, dc, df, da, nullptr, nx, ny, pitch_elems(dtype, ny), hx, hy, omega, -1.0, c, 0, nullptr);
  return down(out, dtype, da, nx, ny);
}

int mg_op_restrict_fw(int in_dtype, int out_dtype, int nx, int ny, const void* fine, void* coarse) {
  CHECK_DEV(valid_dtype(in_dtype) && valid_dtype(out_dtype) && fine && coarse, "mg_op_restrict_fw: bad argument");
  CHECK_DEV(nx >= 3 && ny >= 3 && (nx - 1) % 2 == 0 && (ny - 1) % 2 == 0, "Cannot restrict: fine grid is not coarsenable");   // transfer.py:65-66
  RC(need_device());
  const int cx = (nx - 1) / 2 + 1, cy = (ny - 1) / 2 + 1;
  Scratch s; void *dfine, *dc;
  RC(s.get(&dfine, in_dtype, nx, ny)); RC(s.get(&dc, out_dtype, cx, cy));
  RC(up(dfine, in_dtype, fine, nx, ny));
  d_restrict(in_dtype, out_dtype, dfine, dc, nx, ny, pitch_elems(in_dtype, ny), pitch_elems(out_dtype, cy), nullptr);
  return down(coarse, out_dtype, dc, cx, cy);
}

int mg_op_prolong_bilinear(int in_dtype, int out_dtype, int ncx, int ncy, const void* coarse, void* fine) {
  CHECK_DEV(valid_dtype(in_dtype) && valid_dtype(out_dtype) && coarse && fine && ncx >= 2 && ncy >= 2, "mg_op_prolong_bilinear: bad argument");
  RC(need_device());
  const int nx = 2 * (ncx - 1) + 1, ny = 2 * (ncy - 1) + 1;
  Scratch s; void *dc, *dfine;
  RC(s.get(&dc, in_dtype, ncx, ncy)); RC(s.get(&dfine, out_dtype, nx, ny));
  RC(up(dc, in_dtype, coarse, ncx, ncy));
  // the interpolation runs in the FINE array's dtype (operators/transfer.py:207,236)
  const int rc = d_prolong<false>(in_dtype, out_dtype, (in_dtype == MG_F32 && out_dtype == MG_F32) ? MG_F32 : MG_F64, dc, dfine,
                                  nx, ny, pitch_elems(out_dtype, ny), pitch_elems(in_dtype, ncy), nullptr);
  if (rc != MG_OK) return fail(nullptr, rc, "mg_op_prolong_bilinear: unsupported dtype combination");
  return down(fine, out_dtype, dfine, nx, ny);
}

int mg_op_coarse_solve(int dtype, int nx, int ny, double hx, double hy, double coeff, double tol, int maxit, const void* u0,
                       const void* rhs, void* out, int* sweeps) {
  CHECK_DEV(valid_dtype(dtype) && u0 && rhs && out && nx >= 3 && ny >= 3 && maxit >= 1, "mg_op_coarse_solve: bad argument");
  RC(need_device());
  Scratch s; void *du, *df; void* dsw;
  RC(s.get(&du, dtype, nx, ny)); RC(s.get(&df, dtype, nx, ny));
  RC(alloc_zero(nullptr, &dsw, sizeof(int))); s.ptrs.push_back(dsw);
  RC(up(du, dtype, u0, nx, ny)); RC(up(df, dtype, rhs, nx, ny));
  d_coarse(dtype, du, df, nx, ny, pitch_elems(dtype, ny), hx, hy, coeff, 1.0, tol, maxit, (int*)dsw, nullptr);
  RC(down(out, dtype, du, nx, ny));
  if (sweeps) HIPC(nullptr, hipMemcpy(sweeps, dsw, sizeof(int), hipMemcpyDeviceToHost));
  return MG_OK;
}

}  // extern "C"
