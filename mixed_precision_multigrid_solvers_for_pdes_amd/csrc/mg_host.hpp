// Host-side types shared by the translation units of libmghip.so (mghip.hip: the driver and the C ABI; mg_tail.hip: the
// register-resident coarse tail).  Internal: nothing here is part of the C ABI (include/mghip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/mghip.h"
#include "mg_kernels.hpp"

namespace mgh {

inline size_t esize(int dt) { return dt == MG_F32 ? 4 : 8; }
inline bool valid_dtype(int dt) { return dt == MG_F32 || dt == MG_F64; }

struct Coef {
  double ihx2, ihy2, diag, invD;
  bool pow2;       // 1/diag is exact: multiply instead of divide
  bool all_pow2;   // hx^2, hy^2 and diag are all powers of two
};
// sigma: Helmholtz shift, A = coeff * (Laplacian_h - sigma I) (coeff = -1: -Laplacian + sigma); it only moves the
// diagonal, so every constant-coefficient kernel serves the shifted operator unchanged (sigma = 0: the reference's).
inline Coef coefs(double hx, double hy, double sigma = 0.0) {
  Coef c;
  c.ihx2 = 1.0 / (hx * hx);
  c.ihy2 = 1.0 / (hy * hy);
  c.diag = 2.0 / (hx * hx) + 2.0 / (hy * hy);   // operators/laplacian.py:76, smoothers.py:65
  if (sigma != 0.0) c.diag += sigma;
  c.invD = 1.0 / c.diag;
  int e = 0;
  c.pow2 = std::frexp(c.diag, &e) == 0.5;
  c.all_pow2 = c.pow2 && std::frexp(hx * hx, &e) == 0.5 && std::frexp(hy * hy, &e) == 0.5;
  return c;
}

// min{x >= 0 : sqrt(x) >= tol}: "sqrt(x) < tol" and "x < sqrt_threshold(tol)" decide alike for every double x (IEEE sqrt is
// correctly rounded, hence monotone) -- lets a latency-bound stop test skip the square root.  tol <= 0 never stops.
inline double sqrt_threshold(double tol) {
  if (!(tol > 0.0)) return 0.0;
  thread_local double last_tol = -1.0, last_thr = 0.0;       // one tolerance per solver in practice: launched per tail visit
  if (tol == last_tol) return last_thr;
  last_tol = tol;
  double y = tol * tol;
  while (y > 0.0 && std::sqrt(y) >= tol) y = std::nextafter(y, 0.0);
  while (std::sqrt(y) < tol) y = std::nextafter(y, INFINITY);
  last_thr = y;
  return y;
}

struct Level {
  int nx = 0, ny = 0;
  double hx = 0, hy = 0;
  int ld[2] = {0, 0};
  void* u[2] = {nullptr, nullptr};     // current iterate
  void* t[2] = {nullptr, nullptr};     // Jacobi ping-pong partner (same boundary ring as u)
  void* s[2] = {nullptr, nullptr};     // level 0, allocated on first use: third buffer of the spanning leg (cycle_span)
  void* rhs[2] = {nullptr, nullptr};
  void* r[2] = {nullptr, nullptr};     // residual
  void* a[2] = {nullptr, nullptr};     // diffusion coefficient (variable-coefficient operator), else null
  void* rd[2] = {nullptr, nullptr};    // its reciprocal diagonal 1 / D per cell (var_rdiag_kernel): what the sweeps multiply by
  double timings[3] = {0, 0, 0};       // smooth / restrict / prolong seconds (cfg.profile)
};


// Experiment switches (tile heights, streaming hints, schedule limits) are read from the environment only in measurement
// builds (-DMG_EXPERIMENTS, tools/README.md); the shipped library ignores them.
#ifdef MG_EXPERIMENTS
inline int exp_env(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }
#else
inline int exp_env(const char*, int dflt) { return dflt; }
#endif

}  // namespace mgh

struct mg_handle {
  mg_config cfg;
  std::vector<mgh::Level> lv;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;   // created by mg_create; `stream` may be redirected by mg_set_stream
  double* partials = nullptr;   // device: one fp64 partial per workgroup of the largest reduction (sized in mg_create)
  double* d_scalar = nullptr;   // device, one double
  int* d_int = nullptr;         // device, one int (coarse sweeps)
  double* h_scalar = nullptr;   // pinned host
  int* h_int = nullptr;         // pinned host
  mg::HostMailbox* mbox = nullptr;       // pinned, mapped: the norm of every iteration arrives here
  mg::HostMailbox* mbox_dev = nullptr;   // its device-side address
  unsigned long long mbox_seq = 0;
  void* staging = nullptr;      // fine-level sized fp64 staging for dtype-converting transfers
  int grid_dtype = MG_F64;      // the reference Grid's dtype: MG_F32 only for MG_PREC_SINGLE
  int phase = MG_F64;           // working precision of the adaptive policy
  bool promoted = false;        // one-way rule: fp32 -> fp64 happened
  double fp32_floor = 0.0;      // adaptive policy: eps32 * diag(A) * ||u||_h of the current solve's fp32 phase (0: not evaluated)
  int switch_reason = 0;        // why the fp32 phase of the current solve ended (mg_stats.switch_reason)
  bool have_rhs = false;
  bool varcoef = false;          // A = coeff * div(a grad .) with the per-level fields lv[l].a
  double sigma = 0.0;            // Helmholtz shift: A = coeff * (Laplacian - sigma I) on every level (mg_set_shift)
  double rd_sigma = -1.0;        // the shift the reciprocal diagonals lv[l].rd were computed for (< 0: stale)
  double ring_sumsq[2] = {0, 0};   // sum of f^2 over the boundary ring of the fine rhs, per dtype (r = f there)
  unsigned rhs_gen = 1;            // bumped by every new right-hand side
  unsigned rings_gen[2] = {0, 0};  // rhs_gen the coarse rhs rings of working precision p were injected for (adaptive policy)
  bool iterate_zero = false;       // the fine iterate is zero everywhere (mg_set_solution(NULL) / mg_zero_solution_device, no cycle since)
  unsigned zero_norm_gen[2] = {0, 0};   // ||f - A 0|| = ||f|| as the norm kernel sums it, per dtype, for right-hand side rhs_gen
  double zero_norm_val[2] = {0, 0};
  int norm_partials = 0;           // > 0: `partials` holds sum r^2 over interior cells of the CURRENT fine iterate
  int tail_start = -1;             // first level of the single-workgroup LDS tail (-1: none)
  bool span_ring[2] = {false, false};   // the third level-0 buffer (Level::s) carries the Dirichlet ring of this solve
  int tail2_start = -1;            // first level of the register-resident tail (mg_tail.hip; -1: none); it takes precedence
  int tail2_ntop = 0;              // points per side of that level (65, 33 or 17)
  int* d_tail_ops = nullptr;       // device copy of the tail schedule
  int tail_nops = 0;
  bool tail_direct = false;        // cfg.coarse_direct applies: 5 x 5 coarsest grid inside the tail; tail_minv is its inverse
  double tail_minv[81] = {0};
  double* d_minv = nullptr;        // a coarsest grid other than 5 x 5 with <= 64 unknowns: its n x n inverse on the device (LDS tail)
  int minv_n = 0;
  double tail_minv_sigma = -1.0;   // the shift tail_minv was built for (rebuilt when mg_set_shift changes it)
  std::string err;
  std::vector<double> adapt_hist;

  int L() const { return (int)lv.size(); }
  // precision a level computes in (solvers/multigrid.py:275-285 + core/precision.py:337-357); the coarsest
  // level is never converted by the reference (multigrid.py:270-272 returns first) and stays in the grid dtype.
  int level_dtype_in(int l, int ph) const {
    if (l == L() - 1) return grid_dtype;
    switch (cfg.precision) {
      case MG_PREC_SINGLE: return MG_F32;
      case MG_PREC_SINGLE_MANAGED: return MG_F32;
      case MG_PREC_DEFECT: return MG_F32;          // the error equation's hierarchy; the iterate itself is fp64 (iterate_dtype)
      case MG_PREC_MIXED_LEVELS: return (l >= (cfg.mixed_split > 0 ? cfg.mixed_split : L() / 2)) ? MG_F32 : MG_F64;
      case MG_PREC_ADAPTIVE: return ph;
      default: return MG_F64;
    }
  }
  int level_dtype(int l) const { return level_dtype_in(l, phase); }
  bool fused() const { return cfg.fused != 0 && (cfg.smoother == MG_JACOBI || cfg.smoother == MG_RBGS); }
  bool needs(int l, int dt) const {
    if (cfg.precision == MG_PREC_ADAPTIVE) return (l == L() - 1) ? dt == grid_dtype : true;
    if (cfg.precision == MG_PREC_DEFECT && l == 0 && dt == MG_F64) return true;      // fp64 iterate, its ping-pong partner and f
    return level_dtype(l) == dt;
  }
  // precision of the fine iterate the caller sets / gets: the level-0 working precision, except for defect correction
  int iterate_dtype() const { return cfg.precision == MG_PREC_DEFECT ? MG_F64 : level_dtype(0); }
};


namespace mgh {
// mg_tail.hip: the register-resident coarse tail (mg_tail_kernels.hpp).  tail2_plan decides whether it serves the handle's
// hierarchy (and from which level) and fills h->tail2_*; tail2_launch runs one visit of that sub-cycle on h->stream.
int tail2_plan(mg_handle* h);
int tail2_launch(mg_handle* h, bool zero_top);
}  // namespace mgh
