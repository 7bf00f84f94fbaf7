// Host side of the register-resident coarse tail (mg_tail_kernels.hpp): which hierarchies it serves, and the launch.
#include "mg_host.hpp"
#include "mg_tail_kernels.hpp"

#include <cmath>
#include <cstring>

namespace mgh {

namespace {

// every level from k to L - 2 is a square 2^m + 1 grid on square cells, the coarsest level is 5 x 5
bool dyadic_square_tail(const mg_handle* h, int k) {
  const int L = h->L();
  if (h->lv[L - 1].nx != 5 || h->lv[L - 1].ny != 5) return false;
  for (int l = k; l < L; ++l) {
    const Level& v = h->lv[l];
    if (v.nx != v.ny) return false;
    if (l + 1 < L && v.nx != 2 * (h->lv[l + 1].nx - 1) + 1) return false;
  }
  return true;
}

template <typename T, typename TCO, typename TC, int SM, int NCTOP, bool DIV, bool VAR = false>
int launch_d(mg_handle* h, const mg::Tail2Args& a, const void* rhs, void* u, bool zero_top) {
  using L = mg::T2Lds<T, TCO, NCTOP>;
  auto k = mg::tail2_kernel<T, TCO, TC, SM, NCTOP, DIV, VAR>;
  static bool attr_done = false;                       // per instantiation
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::kTotal) != hipSuccess)
      return MG_ERR_HIP;
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3(1), dim3(mg::T2Geo<NCTOP>::WAVES * 64), L::kTotal, h->stream, (const T*)rhs, (T*)u, a, zero_top ? 1 : 0,
                     h->d_int);
  return MG_OK;
}

// DIV: some level's diagonal has no exact reciprocal (a Helmholtz shift): divide, as the reference does; otherwise every
// level multiplies by the exact 1 / D (bit-identical to the division)
template <typename T, typename TCO, typename TC, int SM, int NCTOP>
int launch_t(mg_handle* h, const mg::Tail2Args& a, const void* rhs, void* u, bool zero_top) {
  if (h->varcoef) {
    // variable coefficients: every sweep multiplies by the stored reciprocal diagonal (no DIV variant); the coefficient
    // registers of a 65^2 top level do not fit beside the iterate (fp64: 236 registers before any temporary; fp32 spills
    // 15..80 of them to scratch), so these hierarchies enter at 33^2 (tail2_plan)
    if constexpr (NCTOP == 64) return MG_ERR_INVALID_VALUE;
    else return launch_d<T, TCO, TC, SM, NCTOP, false, true>(h, a, rhs, u, zero_top);
  }
  bool div = false;
  for (int l = h->tail2_start; l <= h->L() - 2; ++l) div = div || !coefs(h->lv[l].hx, h->lv[l].hy, h->sigma).pow2;
  return div ? launch_d<T, TCO, TC, SM, NCTOP, true>(h, a, rhs, u, zero_top) : launch_d<T, TCO, TC, SM, NCTOP, false>(h, a, rhs, u, zero_top);
}

template <typename T, typename TCO, typename TC, int SM>
int launch_n(mg_handle* h, const mg::Tail2Args& a, const void* rhs, void* u, bool zero_top, int ntop) {
  switch (ntop) {
    case 65: return launch_t<T, TCO, TC, SM, 64>(h, a, rhs, u, zero_top);
    case 33: return launch_t<T, TCO, TC, SM, 32>(h, a, rhs, u, zero_top);
    case 17: return launch_t<T, TCO, TC, SM, 16>(h, a, rhs, u, zero_top);
    default: return MG_ERR_INVALID_VALUE;
  }
}

template <typename T, typename TCO, typename TC>
int launch_sm(mg_handle* h, const mg::Tail2Args& a, const void* rhs, void* u, bool zero_top, int ntop) {
  return h->cfg.smoother == MG_RBGS ? launch_n<T, TCO, TC, mg::kSmRbgs>(h, a, rhs, u, zero_top, ntop)
                                    : launch_n<T, TCO, TC, mg::kSmJacobi>(h, a, rhs, u, zero_top, ntop);
}

}  // namespace

// The register-resident tail serves hierarchies whose last levels are the dyadic squares 65 / 33 / 17 / 9 / 5 in one
// precision (the coarsest level in the grid dtype): every 2^k + 1 grid on a square domain, constant or variable
// coefficient (the latter from 33^2 down: the face means and the reciprocal diagonal live in registers beside the iterate).  Everything else -- rectangles, non-dyadic cell counts, a coarsest grid that is not
// 5 x 5 -- stays with coarse_tail_kernel (plan_tail in mghip.hip).  mg_config.tail: 1 (default) this kernel where it
// applies, 2 the LDS tail only (A/B runs and tests of that kernel).
int tail2_plan(mg_handle* h) {
  h->tail2_start = -1;
  h->tail2_ntop = 0;
  const int L = h->L();
  if (h->cfg.tail != 1 || !h->fused() || L < 3) return MG_OK;
  for (int k = 1; k <= L - 3; ++k) {                    // at least 17^2 -> 9^2 -> 5^2
    const int n = h->lv[k].nx;
    if (n > (h->varcoef ? 33 : 65)) continue;
    if (!dyadic_square_tail(h, k)) return MG_OK;
    bool uniform = true;
    for (int l = k; l <= L - 2; ++l) uniform = uniform && (h->level_dtype_in(l, MG_F64) == h->level_dtype_in(k, MG_F64));
    if (!uniform) continue;
    const double hx = h->lv[k].hx, hy = h->lv[k].hy;
    if (hx != hy) return MG_OK;
    h->tail2_start = k;
    h->tail2_ntop = n;
    return MG_OK;
  }
  return MG_OK;
}

int tail2_launch(mg_handle* h, bool zero_top) {
  const int k = h->tail2_start, L = h->L();
  const int dt = h->level_dtype(k), dco = h->grid_dtype;
  mg::Tail2Args a;
  std::memset(&a, 0, sizeof(a));
  Level& top = h->lv[k];
  a.ld_top = top.ld[dt];
  a.maxit = h->cfg.coarse_maxit;
  a.pre = h->cfg.pre; a.post = h->cfg.post;
  a.colour_offset = h->cfg.colour_offset & 1;
  a.omega = h->cfg.omega; a.coeff = h->cfg.coeff; a.sigma = h->sigma;
  a.tol_x = sqrt_threshold(h->cfg.coarse_tol);
  a.direct = h->tail_direct ? 1 : 0;
  if (a.direct) std::memcpy(a.minv, h->tail_minv, sizeof(a.minv));
  for (int l = k; l <= L - 2; ++l) {
    const Level& v = h->lv[l];
    const Coef c = coefs(v.hx, v.hy, h->sigma);
    mg::Tail2Level& t = a.lv[l - k];
    t.ihx2 = c.ihx2; t.ihy2 = c.ihy2; t.invD = c.invD; t.diag = c.diag;
    int reps = 1;
    if (h->cfg.cycle == MG_CYCLE_W) reps = 2;
    else if (h->cfg.cycle == MG_CYCLE_F) reps = std::max(1, 1 << std::max(0, L - l - 2));
    a.reps[l - k] = reps;
    if (h->varcoef) { a.a_lv[l - k] = v.a[dt]; a.rd_lv[l - k] = v.rd[dt]; a.a_ld[l - k] = v.ld[dt]; }
  }
  {
    const Level& v = h->lv[L - 1];
    const Coef c = coefs(v.hx, v.hy, h->sigma);
    a.hx2_5 = v.hx * v.hx; a.hy2_5 = v.hy * v.hy; a.diag_5 = c.diag; a.hxhy_5 = v.hx * v.hy; a.exact_5 = c.all_pow2 ? 1 : 0;
    a.ring5 = v.rhs[dco]; a.ring5_ld = v.ld[dco];
    if (h->varcoef) {    // hx^2, hy^2 powers of two are all the variable-coefficient solve needs to multiply by reciprocals
      int e = 0;
      a.exact_5 = (std::frexp(a.hx2_5, &e) == 0.5 && std::frexp(a.hy2_5, &e) == 0.5) ? 1 : 0;
      a.a_lv[mg::kT2MaxLev] = v.a[dco]; a.a_ld[mg::kT2MaxLev] = v.ld[dco];
    }
  }
  int rc;
  if (dt == MG_F64) rc = launch_sm<double, double, double>(h, a, top.rhs[dt], top.u[dt], zero_top, h->tail2_ntop);
  else if (dco == MG_F32) rc = launch_sm<float, float, float>(h, a, top.rhs[dt], top.u[dt], zero_top, h->tail2_ntop);
  else rc = launch_sm<float, double, double>(h, a, top.rhs[dt], top.u[dt], zero_top, h->tail2_ntop);
  return rc;
}

}  // namespace mgh

#if MG_EXP_TAIL_TRACE
// timing experiment: the (id, ticks) stamps of the last tail2_kernel launch (mg_tail_kernels.hpp)
extern "C" int mg_exp_tail2_trace(long long* out, int cap) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mg::g_tail2_trace), sizeof(long long) * (size_t)cap) == hipSuccess ? MG_OK : MG_ERR_HIP;
}
#endif
