/*
 * mghip -- C ABI of the MI355X-native geometric-multigrid hot path.
 *
 * This is the drop-in boundary for the reference's V/W-cycle path
 * (Tani843/Mixed_Precision_Multigrid_Solvers_for_PDEs).  The reference is pure
 * Python; the calls a maintainer would re-bind through ctypes are listed on each
 * entry point as "replaces: <file:line>" (paths relative to the reference's
 * src/multigrid/).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every function returns an int status: MG_OK (0) or a negative MG_ERR_*;
 *     the message of the last failure is available from mg_last_error();
 *   - host arrays are dense C-order (nx, ny), j contiguous (core/grid.py:50),
 *     owned by the caller, never retained or modified by the library;
 *   - device arrays (mg_dev_*) are (nx, ny) with a row pitch `ld` in ELEMENTS,
 *     base 16-byte aligned and ld a multiple of 16 bytes (4 f32 / 2 f64);
 *   - calls are blocking unless stated otherwise; a handle is not thread-safe;
 *   - dtype arguments are mg_dtype values; fp32 fields are processed in fp32 and
 *     fp64 fields in fp64, exactly like NumPy does for the reference.
 */
#ifndef MGHIP_H
#define MGHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_ERR_INVALID_VALUE (-1) /* -> ValueError (core/grid.py:34-35,148-149; operators/laplacian.py:61-62; operators/transfer.py:65-69,201-205) */
#define MG_ERR_NO_DEVICE (-2)     /* -> RuntimeError; the reference raises ImportError("CuPy is required") gpu/gpu_solver.py:64-65 */
#define MG_ERR_HIP (-3)           /* -> RuntimeError: a HIP call failed */
#define MG_ERR_STATE (-4)         /* -> ValueError("Multigrid not properly setup or grid mismatch") solvers/multigrid.py:205-206 */
#define MG_ERR_ALLOC (-5)         /* -> MemoryError */

typedef enum { MG_F32 = 0, MG_F64 = 1 } mg_dtype;
typedef enum {
  MG_JACOBI = 0, /* weighted Jacobi            solvers/smoothers.py:41-86, solvers/iterative.py:72-108 */
  MG_RBGS = 1,   /* red-black Gauss-Seidel     solvers/smoothers.py:175-207                          */
  MG_LEXGS = 2   /* lexicographic Gauss-Seidel solvers/smoothers.py:153-173 -- sequential by nature:
                    one workgroup sweeps anti-diagonals; exact, meant for small grids / the coarsest level */
} mg_smoother_t;
typedef enum { MG_CYCLE_V = 0, MG_CYCLE_W = 1, MG_CYCLE_F = 2 } mg_cycle_t;
typedef enum {
  MG_PREC_DOUBLE = 0,       /* Grid(dtype=float64), no precision manager                       */
  MG_PREC_SINGLE = 1,       /* Grid(dtype=float32)                                             */
  MG_PREC_MIXED_LEVELS = 2, /* PrecisionManager('mixed'): level >= L//2 fp32 (core/precision.py:337-357) */
  MG_PREC_ADAPTIVE = 3,     /* threshold switch fp32 <-> fp64 (core/precision.py:270-302)      */
  MG_PREC_SINGLE_MANAGED = 4,/* PrecisionManager('single', adaptive=False) on a float64 Grid: every level converted to
                               fp32 on entry (solvers/multigrid.py:281-285) except the coarsest, which the reference never
                               converts (:270-272) and solves in fp64; interpolation in fp64 (operators/transfer.py:207) */
  MG_PREC_DEFECT = 5        /* defect correction (iterative refinement): fp64 iterate and residual, one fp32 cycle (levels as
                               MG_PREC_SINGLE_MANAGED) from the zero correction on A e = r per outer step, u += e in fp64.
                               The working idea behind gpu/cuda_kernels.py:843-883,937-967 (fp32 iterate, fp64 residual) and
                               :915-929 (correction across precisions); prec_hist code 3.  Constant coefficients only. */
} mg_precision_t;

/* Packed solver configuration: the constructor kwargs of MultigridSolver
 * (solvers/multigrid.py:36-47) / GPUMultigridSolver (gpu/gpu_solver.py:32-46), the Grid
 * (core/grid.py:18-44), LaplacianOperator.coefficient (operators/laplacian.py:22) and the
 * PrecisionManager thresholds (core/precision.py:26-45). */
typedef struct mg_config {
  int32_t nx, ny;            /* fine grid points incl. boundary                                 */
  double x0, x1, y0, y1;     /* domain; hx = (x1-x0)/(nx-1)                                     */
  double coeff;              /* operator A = coeff * Laplacian_h; the consistent choice is -1   */
  int32_t max_levels;        /* hierarchy stops earlier at coarse n < 5 or odd (n-1)            */
  int32_t cycle;             /* mg_cycle_t                                                      */
  int32_t pre, post;         /* smoothing sweeps                                                */
  int32_t smoother;          /* mg_smoother_t                                                   */
  double omega;              /* relaxation parameter                                            */
  double coarse_tol;         /* coarsest lex-GS: stop at ||r|| < coarse_tol ...                 */
  int32_t coarse_maxit;      /* ... or after this many sweeps                                   */
  int32_t precision;         /* mg_precision_t                                                  */
  double switch_threshold;   /* PrecisionManager.convergence_threshold                          */
  double memory_threshold_gb;/* PrecisionManager.memory_threshold_gb                            */
  int32_t adaptive_reference_rule; /* 1: the reference's two-way rule verbatim (never recovers);
                                      0: one-way fp32 -> fp64 at ||r|| < 10*thr, on fp32 stagnation, or as soon as ||r|| is
                                      within 2 x of the fp32 residual floor eps32 * diag(A) * ||u||_h (evaluated once, after
                                      the first fp32 cycle: more fp32 cycles cannot lower the residual any further) */
  int32_t device;            /* HIP device ordinal                                              */
  int32_t profile;           /* 1: per-level stage timings (synchronising; solvers/multigrid.py:179-182) */
  int32_t colour_offset;     /* parity of the global index of local cell (0,0) (sub-domains)     */
  int32_t fused;             /* 0: one launch per operator; 1: fused legs (sweeps+residual+restriction / prolongation+sweeps+norm:
                                two launches per level, identical arithmetic) with the iterate tiled through LDS; 2: the same
                                legs register-blocked (iterate in registers, DPP lateral neighbours) on levels above ~1100^2
                                cells, LDS-tiled below; 3: register-blocked on every level (tests) */
  int32_t tail;              /* with fused >= 1 -- 1: the coarse levels incl. the coarsest solve run in ONE workgroup per visit: the
                                dyadic square levels 65^2 / 33^2 / 17^2 / 9^2 / 5^2 of a constant-coefficient hierarchy with
                                iterate and rhs in REGISTERS (csrc/mg_tail_kernels.hpp), any other <= ~33^2 (fp64) / ~65^2 (fp32)
                                sub-hierarchy with its fields in LDS; 2: the LDS kernel only; 0: per-level launches.  Same
                                arithmetic per cell in all three */
  int32_t fmg_cycles;        /* > 0: mg_solve without an initial guess starts from a full-multigrid guess with this many
                                cycles per level (solvers/advanced_multigrid.py:626-683, gpu/gpu_solver.py:583-652) */
  int32_t speculate;         /* with fused -- 1: mg_iterate / mg_solve queue the down leg of cycle k+1 while ||r_k|| travels to
                                the host (dropped if that norm ends the solve); 2 (the host side's default): as 1, and where the
                                finest level is bandwidth-bound (> ~1100^2 cells, constant coefficients) the up leg of
                                cycle k and that down leg are ONE launch (the spanning leg, csrc/mg_rb_kernels.hpp): the iterate
                                between the two cycles is written (unless tol <= 0 and no precision switch is pending: nothing
                                can end the solve there) but never read back.  Same iterates bit for bit; the norm's partial
                                sums are taken over other tiles (last-bit differences).  0: strictly one cycle at a time */
  int32_t coarse_direct;     /* 1 (with fused and tail, a coarsest grid of at most 64 unknowns: the 5 x 5 of every 2^k + 1 square --
                                nine --, the 9 x 5 of a 2:1 domain -- 21 --, ...): the coarsest system is solved
                                directly (u = A^-1 f, the inverse formed on the host) instead of by the reference's Gauss-Seidel
                                iteration to coarse_tol (solvers/multigrid.py:119-124, 355-370).  NOT bit-identical to the
                                reference: the two differ by at most ||A_c^-1|| coarse_tol / h_c ~ 2e-13 per coarsest visit (the
                                error the iteration is allowed to leave; iterates stay within 1e-12 relative l-inf of the
                                reference's).  0: always the iteration (bit-identical; what the parity tests
                                run).  < 0 (the host side's default): the same as 1 */
  int32_t mixed_split;       /* MG_PREC_MIXED_LEVELS: first fp32 level; <= 0: num_levels / 2 (core/precision.py:351-357).
                                Set by a caller whose handle is the lower part of a longer hierarchy (distributed.py: the
                                replicated coarse levels below the decomposed ones keep the GLOBAL split) */
} mg_config;

typedef struct mg_stats {
  double solve_seconds;      /* device-resident cycles + norms                (gpu_solve_time)  */
  double h2d_seconds;        /* rhs / initial guess upload                    (gpu_transfer_time, part) */
  double d2h_seconds;        /* solution download                                               */
  double initial_residual;   /* ||f - A u0||                                  (gpu/gpu_solver.py:243-251) */
  int32_t precision_switches;
  int32_t last_coarse_sweeps;
  int32_t switch_reason;     /* MG_PREC_ADAPTIVE: why the fp32 phase ended -- 0 it did not (or never began), 1 ||r|| < 10 x
                                switch_threshold (core/precision.py:248-268), 2 stagnation (core/precision.py:189-246),
                                3 ||r|| within 2 x of the fp32 residual floor eps32 * diag(A) * ||u|| (ours), 4 the fp32 phase
                                was never entered: that floor, bounded a priori by eps32 diag(A) / lambda_min ||r_0||, leaves it
                                fewer than two useful cycles (ours; the solve is then a double solve) */
  double fp32_floor;         /* that floor estimate (0: not evaluated) */
} mg_stats;

typedef struct mg_handle mg_handle;

/* ---- library / device ------------------------------------------------------------------ */
const char* mg_version(void);
int mg_device_count(int* count);
/* message of the last error on this handle (NULL: last error of a handle-less call). */
const char* mg_last_error(const mg_handle* h);

/* ---- solver object: replaces GPUMultigridSolver.setup/solve/cleanup ---------------------- */
/* replaces: gpu/gpu_solver.py:116-184 (setup: hierarchy + per-level device arrays), solvers/multigrid.py:135-182 */
int mg_create(const mg_config* cfg, mg_handle** out);
/* replaces: gpu/gpu_solver.py:483-501 (cleanup) */
int mg_destroy(mg_handle* h);
int mg_num_levels(const mg_handle* h, int* n);
int mg_level_shape(const mg_handle* h, int level, int* nx, int* ny);
/* accumulated seconds {smooth, restrict, prolong} of `level` when cfg.profile = 1 (solvers/multigrid.py:289-335) */
int mg_level_timings(const mg_handle* h, int level, double out3[3]);

/* replaces: gpu/gpu_solver.py:186-328 and solvers/multigrid.py:184-251.
 * rhs, u0 (nullable) and u_out are host arrays of host_dtype.  hist receives one ||r|| per cycle
 * (at most hist_cap), prec_hist (nullable) the working precision of each cycle (mg_dtype, or 2 for
 * per-level mixed).  Stops when ||r|| < tol (absolute, solvers/base.py:134) or after max_iter cycles. */
int mg_solve(mg_handle* h, const void* rhs, const void* u0, void* u_out, int host_dtype, double tol,
             int max_iter, double* hist, int hist_cap, int* n_iter, int* converged, int32_t* prec_hist,
             mg_stats* stats);

/* The iteration loop of mg_solve alone, on the rhs / iterate already resident on the device
 * (mg_set_rhs, mg_set_solution): policy check, cycle, ||r||, repeated; no host transfer of fields.
 * replaces: the loop body of solvers/multigrid.py:219-246 / gpu/gpu_solver.py:254-297 */
int mg_iterate(mg_handle* h, double tol, int max_iter, double* hist, int hist_cap, int* n_iter, int* converged,
               int32_t* prec_hist, mg_stats* stats);

/* Variable-coefficient operator A = coeff * div(a grad .) (BASELINE config 5; NOT in the reference, SURVEY F12: our
 * design, parity unpinned -- a == 1 reproduces the constant-coefficient path bit for bit on dyadic grids).
 * `a` holds vertex values on the fine grid (host array, (nx, ny)); face values are arithmetic means; coarse
 * operators are re-discretised with `a` injected.  NULL switches back to the constant-coefficient operator.
 * Coarse levels take every 2^l-th vertex value, cast once from the caller's dtype.  Cycles run as fused legs too
 * (coefficient tile staged with the iterate, face means kept in registers: 4.25 words / DoF per leg); fused = 0 keeps
 * one launch per operator (4 words / DoF per sweep). */
int mg_set_coefficient(mg_handle* h, const void* a_host_or_null, int host_dtype);

/* Helmholtz shift: the operator of every level becomes A = coeff * (Laplacian_h - sigma I), i.e. -Laplacian + sigma
 * for coeff = -1 (with mg_set_coefficient: -div(a grad .) + sigma); sigma >= 0, 0 restores the reference's operator.
 * The shift only moves the stencil diagonal, so all kernels (fused legs and LDS tail included) serve it unchanged.
 * This is the linear system of an implicit heat-equation step, (-Laplacian + 1/(alpha dt)) u = rhs / (alpha dt):
 * applications/heat_equation.py:209-220, 254-261 set it up and then relax it with plain Gauss-Seidel
 * (_solve_helmholtz, :459-497) because the reference's multigrid cannot shift its operator. */
int mg_set_shift(mg_handle* h, double sigma);

/* Device-resident stepping (benchmarks, preconditioner-style callers: fixed cycle counts, no transfer). */
int mg_set_rhs(mg_handle* h, const void* rhs, int host_dtype);
int mg_set_solution(mg_handle* h, const void* u0_or_null, int host_dtype);
int mg_get_solution(mg_handle* h, void* u_out, int host_dtype);
int mg_cycle(mg_handle* h, int ncycles);           /* asynchronous on the handle's stream        */
/* full-multigrid initial guess from the resident rhs (replaces solvers/advanced_multigrid.py:626-683), asynchronous */
int mg_fmg(mg_handle* h, int cycles_per_level);
int mg_residual_norm(mg_handle* h, double* out);   /* sqrt(hx*hy*sum r^2), synchronises           */
int mg_set_working_precision(mg_handle* h, int dtype); /* MG_PREC_ADAPTIVE only: in-device cast of u */
int mg_synchronize(mg_handle* h);
/* Queue the handle's work on the caller's stream (use_own = 0) or back on its own (use_own = 1). */
int mg_set_stream(mg_handle* h, void* stream, int use_own);
/* Device-to-device forms of mg_set_rhs / mg_set_solution(NULL) / mg_get_solution: (nx, ny) arrays with pitch
 * `ld`, converted to/from the handle's working precision on the handle's stream, asynchronous. */
int mg_set_rhs_device(mg_handle* h, const void* rhs_dev, int ld, int dtype);
/* the same, for a right-hand side whose BOUNDARY RING equals that of the last mg_set_rhs / mg_set_rhs_device (the coarse
 * right-hand side of a decomposed cycle: its ring is the injected ring of f, cycle after cycle): the rings of the coarser
 * levels and their sums are kept instead of being injected again */
int mg_update_rhs_device(mg_handle* h, const void* rhs_dev, int ld, int dtype);
int mg_zero_solution_device(mg_handle* h);
int mg_get_solution_device(mg_handle* h, void* u_dev, int ld, int dtype);
/* the stream all of the handle's work is queued on (a hipStream_t), for callers that bracket with events */
int mg_get_stream(mg_handle* h, void** stream);

/* hipEvent-timed repetitions of one kernel of the path on the handle's own arrays and stream
 * (used by bench.py for the roofline line).  op: 0 jacobi sweep, 1 rbgs sweep (both colours),
 * 2 residual (store r), 3 residual+norm (no store), 4 restrict, 5 prolong+add, 6 whole cycle,
 * 7 fused down leg (2 sweeps + residual + restriction), 8 fused up leg (prolongation + 2 sweeps [+ norm on level 0]),
 * 9 two fused sweeps, 10 the op-0 Jacobi sweep rotating over >= 3 independent {u, rhs, out} sets of > 768 MiB in total
 * (allocated for the call), so that no launch finds its operands in the 256 MiB Infinity Cache: the HBM-proper figure;
 * 11 a bare c = a + b stream over the same rotating sets (the memory system's ceiling for the sweep's 2-read + 1-write shape).
 * dtype selects the precision of `level`'s arrays (must be allocated under cfg.precision). */
int mg_time_op(mg_handle* h, int op, int level, int dtype, int reps, double* avg_ms);

/* ---- stateless operators on HOST arrays (upload, run the HIP kernel, download) ----------- */
/* replaces: operators/laplacian.py:105-124 / gpu/cuda_kernels.py:794-828 (TransferKernels.compute_residual) */
int mg_op_residual(int dtype, int nx, int ny, double hx, double hy, double coeff, const void* u, const void* f, void* r);
/* replaces: gpu/cuda_kernels.py:843-883, 937-967 (MixedPrecisionKernels.compute_mixed_precision_residual): fp32 iterate and
 * rhs in, fp64 residual out, evaluated in double (16 B / DoF) -- with the operator and the boundary convention of
 * mg_op_residual (A = coeff * Laplacian_h, r = f on boundary cells) in place of that kernel's own. */
int mg_op_residual_mixed(int nx, int ny, double hx, double hy, double coeff, const float* u, const float* f, double* r);
/* replaces: operators/laplacian.py:44-80 (apply = f - residual with f = 0, sign folded) */
int mg_op_apply(int dtype, int nx, int ny, double hx, double hy, double coeff, const void* u, void* au);
/* replaces: core/grid.py:174-187 (Grid.l2_norm) */
int mg_op_norm(int dtype, int nx, int ny, double hx, double hy, const void* field, double* out);
/* replaces: solvers/smoothers.py:41-86, solvers/iterative.py:72-108, gpu/cuda_kernels.py:284-346 (jacobi_smoothing) */
int mg_op_jacobi(int dtype, int nx, int ny, double hx, double hy, double omega, int nu, const void* u, const void* rhs, void* out);
/* replaces: solvers/smoothers.py:117-151,175-207, gpu/cuda_kernels.py:348-390 (red_black_gauss_seidel) */
int mg_op_rbgs(int dtype, int nx, int ny, double hx, double hy, double omega, int nu, const void* u, const void* rhs, void* out);
/* shifted operator on host arrays -- op 0: out = f - A_sigma u; 1: nu weighted-Jacobi sweeps; 2: nu red-black GS sweeps
 * (the sweeps divide by 2/hx^2 + 2/hy^2 + sigma).  See mg_set_shift. */
int mg_op_helmholtz(int dtype, int op, int nx, int ny, double hx, double hy, double coeff, double sigma, double omega, int nu,
                    const void* u, const void* f, void* out);
/* variable-coefficient forms of residual / Jacobi / red-black GS (no reference counterpart, see mg_set_coefficient) */
int mg_op_residual_var(int dtype, int nx, int ny, double hx, double hy, double coeff, const void* a, const void* u, const void* f, void* r);
int mg_op_jacobi_var(int dtype, int nx, int ny, double hx, double hy, double omega, int nu, const void* a, const void* u, const void* rhs, void* out);
int mg_op_rbgs_var(int dtype, int nx, int ny, double hx, double hy, double omega, int nu, const void* a, const void* u, const void* rhs, void* out);
/* replaces: operators/transfer.py:53-81,100-124, gpu/cuda_kernels.py:738-764 (TransferKernels.restriction) */
int mg_op_restrict_fw(int in_dtype, int out_dtype, int nx, int ny, const void* fine, void* coarse);
/* replaces: operators/transfer.py:189-215,234-267, gpu/cuda_kernels.py:766-792 (TransferKernels.prolongation) */
int mg_op_prolong_bilinear(int in_dtype, int out_dtype, int ncx, int ncy, const void* coarse, void* fine);
/* replaces: solvers/smoothers.py:153-173 under solvers/base.py:234-290 (coarsest-grid solve) */
int mg_op_coarse_solve(int dtype, int nx, int ny, double hx, double hy, double coeff, double tol, int maxit,
                       const void* u0, const void* rhs, void* out, int* sweeps);

/* ---- stateless operators on DEVICE arrays (asynchronous on `stream`, a hipStream_t or NULL) -- */
int mg_dev_jacobi(int dtype, int nx, int ny, int ld, double hx, double hy, double omega,
                  const void* u, const void* rhs, void* out, void* stream);
int mg_dev_rbgs_colour(int dtype, int nx, int ny, int ld, double hx, double hy, double omega, int colour,
                       int colour_offset, void* u, const void* rhs, void* stream);
int mg_dev_residual(int dtype, int nx, int ny, int ld, double hx, double hy, double coeff,
                    const void* u, const void* f, void* r, void* stream);
/* device form of mg_op_residual_mixed: u, f fp32 with pitch ld_in, r fp64 with pitch ld_out (elements) */
int mg_dev_residual_f32in_f64out(int nx, int ny, int ld_in, int ld_out, double hx, double hy, double coeff, const float* u,
                                 const float* f, double* r, void* stream);
/* sum of squares of field[i_lo:i_hi, j_lo:j_hi] into *sumsq_dev (one double in device memory);
 * scratch >= mg_dev_scratch_bytes().  The window lets a sub-domain count the cells it owns. */
int mg_dev_sumsq(int dtype, int ld, int i_lo, int i_hi, int j_lo, int j_hi, const void* field, void* scratch,
                 double* sumsq_dev, void* stream);
/* Sub-domain forms of the transfers.  Coarse cell (ic, jc) sits on fine cell (2ic, 2jc).  `sides` is a bit
 * mask of the edges of THIS array that are physical boundaries (1: i = 0, 2: i = n-1, 4: j = 0, 8: j = n-1;
 * 15 for a whole grid); the remaining edges are ghost rings owned by a neighbouring sub-domain: restriction
 * leaves those coarse cells untouched, prolongation interpolates them from the coarse ghost values. */
int mg_dev_restrict_fw(int in_dtype, int out_dtype, int nxf, int nyf, int ldf, int nxc, int nyc, int ldc, int sides,
                       const void* fine, void* coarse, void* stream);
int mg_dev_prolong_add(int coarse_dtype, int fine_dtype, int compute_dtype, int nxf, int nyf, int ldf, int nxc,
                       int nyc, int ldc, int sides, const void* coarse, void* fine_u, void* stream);
int mg_dev_convert(int in_dtype, int out_dtype, int nx, int ny, int ldi, int ldo, const void* in, void* out, void* stream);
/* Fused legs on device arrays (the kernels of the single-GPU engine).  On a sub-domain the local
 * array carries a ghost zone several cells wide: every edge is treated as fixed, so after s sweeps the outer s cells
 * of a ghost zone are stale -- the caller sizes the zone so that the cells it owns stay exact.  Coarse cell (ic, jc)
 * sits on fine cell (2 (ic - ci_off), 2 (jc - cj_off)).
 *   down leg: nsweep (<= 2) sweeps of u (or of the zero iterate: zero_init) -> out; residual; full weighting into the
 *             interior cells of rhs_coarse that have a complete fine neighbourhood here.  `select` = 0 runs every tile;
 *             1 only the tiles whose staged region lies inside inner_rect = {i_lo, i_hi, j_lo, j_hi} (cells that need no
 *             ghost data: launch them while the halo exchange is in flight), 2 only the remaining tiles.
 *   up leg:   out = sweeps(u + P e_coarse); with norm != 0 also *sumsq_dev = sum of r^2 over cells
 *             [ni_lo, ni_hi) x [nj_lo, nj_hi) that are interior to this array (scratch >= mg_dev_scratch_bytes()). */
int mg_dev_down_leg(int smoother, int dtype, int coarse_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc, int ci_off,
                    int cj_off, double hx, double hy, double omega, double coeff, int nsweep, int zero_init, int colour_offset,
                    const void* u, const void* rhs, void* out, void* rhs_coarse, void* stream, int select, const int* inner_rect);
int mg_dev_up_leg(int smoother, int dtype, int coarse_dtype, int compute_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc,
                  int ci_off, int cj_off, int sides, double hx, double hy, double omega, double coeff, int nsweep, int colour_offset,
                  const void* u, const void* rhs, void* out, const void* e_coarse, int norm, int ni_lo, int ni_hi, int nj_lo,
                  int nj_hi, void* scratch, double* sumsq_dev, void* stream);
/* The same legs for the variable-coefficient operator A = coeff * div(a grad .): `acoef` holds the vertex values of a on
 * this array (dtype / pitch of u, ghost zone included; NULL = the constant-coefficient legs above) and `rdiag` its
 * reciprocal diagonal per cell, computed once per coefficient by mg_dev_var_rdiag (same dtype / pitch; NULL with acoef NULL):
 * the sweeps multiply by it, no division per cell and sweep.  No reference counterpart; see mg_set_coefficient. */
int mg_dev_down_leg_var(int smoother, int dtype, int coarse_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc, int ci_off,
                        int cj_off, double hx, double hy, double omega, double coeff, int nsweep, int zero_init, int colour_offset,
                        const void* u, const void* rhs, void* out, void* rhs_coarse, void* stream, int select, const int* inner_rect,
                        const void* acoef, const void* rdiag);
int mg_dev_up_leg_var(int smoother, int dtype, int coarse_dtype, int compute_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc,
                      int ci_off, int cj_off, int sides, double hx, double hy, double omega, double coeff, int nsweep, int colour_offset,
                      const void* u, const void* rhs, void* out, const void* e_coarse, int norm, int ni_lo, int ni_hi, int nj_lo,
                      int nj_hi, void* scratch, double* sumsq_dev, void* stream, const void* acoef, const void* rdiag);
/* The spanning leg (csrc/mg_rb_kernels.hpp, rb_span_kernel): mg_dev_up_leg of cycle k (u += P e_coarse, nsweep_post sweeps,
 * sum r^2 over the window) and mg_dev_down_leg of cycle k + 1 (nsweep_pre sweeps, residual, full weighting into rhs_coarse) in
 * ONE pass over `u`: the iterate of cycle k goes to `out_mid` (may be NULL when nobody will read it), the pre-smoothed iterate of
 * cycle k + 1 to `out_next`; u, out_mid, out_next are three different arrays of one shape.  Same bits as the two calls.
 * Weighted Jacobi, constant coefficients, fine and coarse field of one dtype, arrays above ~1100^2 cells (what the
 * register-blocked legs serve); anything else returns MG_ERR_INVALID_VALUE and the caller issues the two legs.
 * mg_dev_span_leg_ok: 1 where the call would be accepted.  Replaces nothing upstream (the reference smooths, restricts and
 * interpolates one operator at a time, solvers/multigrid.py:289-335). */
int mg_dev_span_leg_ok(int smoother, int dtype, int coarse_dtype, int compute_dtype, int nx, int ny);
int mg_dev_span_leg(int smoother, int dtype, int coarse_dtype, int compute_dtype, int nx, int ny, int ld, int nxc, int nyc, int ldc,
                    int ci_off, int cj_off, int sides, double hx, double hy, double omega, double coeff, int nsweep_post,
                    int nsweep_pre, int colour_offset, const void* u, const void* rhs, void* out_mid, void* out_next,
                    const void* e_coarse, void* rhs_coarse, int ni_lo, int ni_hi, int nj_lo, int nj_hi, void* scratch,
                    double* sumsq_dev, void* stream);
/* rdiag[i][j] = 1 / ((a(i+1/2) + a(i-1/2)) / hx^2 + (a(j+1/2) + a(j-1/2)) / hy^2 [+ sigma]) on interior cells (face values =
 * arithmetic means of the vertex values), 0 on the ring of the array; rounded once in `dtype` */
int mg_dev_var_rdiag(int dtype, int nx, int ny, int ld, double hx, double hy, double sigma, const void* a, void* rdiag, void* stream);
/* boundary ring of a coarse field := injected fine values, on the physical edges (`sides`) only */
int mg_dev_inject_ring(int in_dtype, int out_dtype, int nxf, int nyf, int ldf, int nxc, int nyc, int ldc, int sides, int ci_off,
                       int cj_off, const void* fine, void* coarse, void* stream);
int mg_dev_scratch_bytes(int nx, int ny, int64_t* bytes);
/* pitch (elements) the library itself uses for an (nx, ny) field of `dtype` */
int mg_pitch_elems(int dtype, int ny, int* ld);

/* ------------------------------------------------------------------------------------------------
 * Cycle plans: one decomposed V/W/F cycle of one rank as a flat list of operations, replayed natively.
 * The reference's counterpart is the per-cycle Python of its (never-running) DistributedMultigridSolver /
 * halo exchange (gpu/multi_gpu.py:380-460); here the host side records the list once -- pointers, shapes,
 * peers and stream assignment of everything `DistributedMultigrid.cycle()` issues -- and every later cycle is ONE call:
 * kernel launches, halo copies, RCCL send/recv groups, the coarse gather, the replicated engine's cycle and the norm
 * all-reduce are enqueued from C++ on two HIP streams.  All pointers are device pointers owned by the caller and must
 * outlive the plan.  Slots not listed for an operation are ignored.
 * ------------------------------------------------------------------------------------------------ */
enum {
  MG_PLAN_DOWN_LEG = 1,    /* i: smoother, dtype, coarse_dtype, nx, ny, ld, nxc, nyc, ldc, ci_off, cj_off, nsweep, zero_init,
                                 colour_offset, select, has_rect, rect[4];  d: hx, hy, omega, coeff;
                                 p: u, rhs, out, rhs_coarse, acoef, rdiag                            (mg_dev_down_leg_var) */
  MG_PLAN_UP_LEG = 2,      /* i: smoother, dtype, coarse_dtype, compute_dtype, nx, ny, ld, nxc, nyc, ldc, ci_off, cj_off, sides,
                                 nsweep, colour_offset, norm, window[4];  d: hx, hy, omega, coeff;
                                 p: u, rhs, out, e_coarse, scratch, sumsq_dev, acoef, rdiag          (mg_dev_up_leg_var) */
  MG_PLAN_COPY2D = 3,      /* i: rows, width_bytes, dst_pitch_bytes, src_pitch_bytes;  p: dst, src   (4-byte granularity) */
  MG_PLAN_ADD_F64 = 4,     /* p: dst, a, b:  *dst = *a + *b   (device doubles; b NULL: *dst = *a) */
  MG_PLAN_GROUP_BEGIN = 5, /* ncclGroupStart */
  MG_PLAN_SEND = 6,        /* i: peer, bytes;  p: buffer */
  MG_PLAN_RECV = 7,        /* i: peer, bytes;  p: buffer */
  MG_PLAN_GROUP_END = 8,   /* ncclGroupEnd */
  MG_PLAN_ALLGATHER = 9,   /* i: bytes per rank;  p: send, recv */
  MG_PLAN_ALLREDUCE_F64 = 10, /* i: count;  p: buffer (in place, sum) */
  MG_PLAN_COARSE_BEGIN = 11,  /* i: ld, dtype, same_ring (mg_update_rhs_device);  p: engine handle, rhs   (stream, rhs, zero iterate) */
  MG_PLAN_COARSE_CYCLE = 12,  /* i: cycles;  p: engine handle */
  MG_PLAN_COARSE_END = 13,    /* i: ld, dtype;  p: engine handle, out */
  MG_PLAN_EVENT_RECORD = 14,  /* i: event id (0..7) on the operation's stream */
  MG_PLAN_STREAM_WAIT = 15,   /* i: event id: the operation's stream waits for it */
  MG_PLAN_RESULT = 16,        /* p: device double copied to the host at the end of mg_plan_run (at most one) */
  MG_PLAN_SPAN_LEG = 17       /* mg_dev_span_leg -- i: smoother, dtype, coarse_dtype, compute_dtype, nx, ny, ld, nxc, nyc, ldc, ci_off,
                                 cj_off, sides, nsweep_post, nsweep_pre, colour_offset, ni_lo, ni_hi, nj_lo, nj_hi;  d: hx, hy, omega, coeff;
                                 p: u, rhs, out_mid (may be NULL), out_next, e_coarse, rhs_coarse, scratch, sumsq_dev */
};
typedef struct mg_plan_op {
  int32_t op;       /* MG_PLAN_* */
  int32_t stream;   /* 0: compute stream, 1: communication stream */
  int32_t i[24];
  double d[4];
  void* p[8];
} mg_plan_op;
typedef struct mg_plan mg_plan;
/* comm: an RCCL communicator from mg_comm_init, or NULL for a plan without SEND/RECV/ALLGATHER/ALLREDUCE operations */
int mg_plan_create(const mg_plan_op* ops, int n_ops, void* comm, int device, mg_plan** out);
/* enqueue every operation; when the plan has a RESULT, wait for it and store it in *result */
int mg_plan_run(mg_plan* plan, void* compute_stream, void* comm_stream, double* result);
/* the two halves of mg_plan_run: enqueue (the RESULT travels to the host behind an event) / collect the RESULT.  Between them
 * the caller may enqueue more work -- the front part of the next cycle -- so that the device does not wait for the host. */
int mg_plan_run_async(mg_plan* plan, void* compute_stream, void* comm_stream);
int mg_plan_wait(mg_plan* plan, double* result);
int mg_plan_num_ops(const mg_plan* plan, int* n);
/* COPY2D operations of the plan and the launches they run as (runs of independent copies share one launch) */
int mg_plan_copy_launches(const mg_plan* plan, int* n_copies, int* n_launches);
/* Per-phase device times of a plan (diagnostics, outside any timed region): while enabled, every operation of mg_plan_run /
 * mg_plan_run_async is bracketed by a pair of timing events on the stream it runs on.  mg_plan_phase_times waits for the
 * recorded work and ADDS the elapsed milliseconds per phase to out[MG_PLAN_PHASES], then forgets the events:
 *   [0] fused legs  [1] halo pack / unpack / scatter copies  [2] send/recv groups (includes the wait for the peers)
 *   [3] coarse all-gather  [4] replicated coarse engine  [5] norm adds + all-reduce  [6] anything else */
#define MG_PLAN_PHASES 7
int mg_plan_profile(mg_plan* plan, int enable);
int mg_plan_phase_times(mg_plan* plan, double* out_ms);
/* MG_ERR_TIMEOUT from mg_plan_wait / mg_plan_run: the RESULT did not arrive within MG_PLAN_TIMEOUT_S seconds (environment,
 * default 120; plans with a communicator, and any plan when the variable is set).  The queued work is NOT cancelled: the
 * caller must not synchronise on those streams again -- a rank in this state reports and leaves the process. */
#define MG_ERR_TIMEOUT (-6)
const char* mg_plan_error(const mg_plan* plan);   /* NULL plan: the last mg_comm_* / mg_plan_create error of this thread */
int mg_plan_destroy(mg_plan* plan);
/* RCCL through the library the process already uses (`rccl_library`: path of librccl.so, e.g. torch's own copy).
 * mg_comm_unique_id fills 128 bytes on one rank; every rank passes the same bytes to mg_comm_init. */
int mg_comm_unique_id(const char* rccl_library, void* id128);
int mg_comm_init(const char* rccl_library, const void* id128, int nranks, int rank, int device, void** comm);
int mg_comm_destroy(void* comm);
/* ranks of the communicator / this process's rank in it (what the bench line reports as ranks_seen) */
int mg_comm_ranks(void* comm, int* nranks, int* rank);

#ifdef __cplusplus
}
#endif
#endif /* MGHIP_H */
