/*
 * CPU oracle, C restatement -- TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/mg_oracle.py for the NumPy twin,
 * which is the one pinned against the reference's golden vectors; tests/test_oracle_golden.py checks that this
 * file reproduces the NumPy oracle bit for bit).  Used by bench.py's cpu_baseline leg as the multi-threaded
 * "port" of the reference's CPU V-cycle; nothing in the product package links or loads it.
 *
 * fp64 V/W-cycle of the oracle configuration: A = coeff * Laplacian_h (coeff = -1),
 * weighted Jacobi (vectorised form, solvers/iterative.py:84-104) or red-black GS (solvers/smoothers.py:175-207),
 * residual (operators/laplacian.py:73-77,117-118), full weighting with boundary injection
 * (operators/transfer.py:100-124), bilinear prolongation with the far-edge zeros (operators/transfer.py:234-267),
 * coarsest level by lexicographic GS to tol / maxit (solvers/smoothers.py:153-173, solvers/base.py:255-290),
 * cycle recursion of solvers/multigrid.py:253-337.  Arrays are dense (nx, ny), C order.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).  No fast-math: the association order
 * of every expression is the reference's.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(i, j) ((size_t)(i) * ny + (j))

int mgo_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* cap the OpenMP team (bench.py: the GPU box grants one GPU's share of the host cores, not all of them) */
void mgo_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

void mgo_jacobi(const double* u, const double* f, double* out, int nx, int ny, double hx, double hy, double omega) {
  const double hx2_inv = 1.0 / (hx * hx), hy2_inv = 1.0 / (hy * hy);
  const double diag = -(2.0 * hx2_inv + 2.0 * hy2_inv);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ny; ++j) {
      if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) { out[IDX(i, j)] = u[IDX(i, j)]; continue; }
      const double nb = hx2_inv * (u[IDX(i + 1, j)] + u[IDX(i - 1, j)]) + hy2_inv * (u[IDX(i, j + 1)] + u[IDX(i, j - 1)]);
      const double un = (f[IDX(i, j)] + nb) / (-diag);
      out[IDX(i, j)] = (1.0 - omega) * u[IDX(i, j)] + omega * un;
    }
}

void mgo_rbgs_colour(double* u, const double* f, int nx, int ny, double hx, double hy, double omega, int colour) {
  const double hx2 = hx * hx, hy2 = hy * hy;
  const double diag = -2.0 / hx2 - 2.0 / hy2;
#pragma omp parallel for schedule(static)
  for (int i = 1; i < nx - 1; ++i)
    for (int j = 1; j < ny - 1; ++j) {
      if (((i + j) & 1) != colour) continue;
      const double nb = (u[IDX(i + 1, j)] + u[IDX(i - 1, j)]) / hx2 + (u[IDX(i, j + 1)] + u[IDX(i, j - 1)]) / hy2;
      const double un = (f[IDX(i, j)] + nb) / (-diag);
      u[IDX(i, j)] = (1 - omega) * u[IDX(i, j)] + omega * un;
    }
}

void mgo_residual(const double* u, const double* f, double* r, int nx, int ny, double hx, double hy, double coeff) {
  const double hx2 = hx * hx, hy2 = hy * hy;
  const double d = 2.0 / hx2 + 2.0 / hy2;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ny; ++j) {
      if (i == 0 || i == nx - 1 || j == 0 || j == ny - 1) { r[IDX(i, j)] = f[IDX(i, j)]; continue; }
      const double au = coeff * (((u[IDX(i + 1, j)] + u[IDX(i - 1, j)]) / hx2 + (u[IDX(i, j + 1)] + u[IDX(i, j - 1)]) / hy2) -
                                 u[IDX(i, j)] * d);
      r[IDX(i, j)] = f[IDX(i, j)] - au;
    }
}

double mgo_norm(const double* r, int nx, int ny, double hx, double hy) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ny; ++j) s += r[IDX(i, j)] * r[IDX(i, j)];
  return sqrt(hx * hy * s);
}

void mgo_restrict_fw(const double* fine, double* coarse, int nx, int ny) {
  const int cnx = (nx - 1) / 2 + 1, cny = (ny - 1) / 2 + 1;
#pragma omp parallel for schedule(static)
  for (int ic = 0; ic < cnx; ++ic)
    for (int jc = 0; jc < cny; ++jc) {
      const int i = 2 * ic, j = 2 * jc;
      if (ic == 0 || ic == cnx - 1 || jc == 0 || jc == cny - 1) { coarse[(size_t)ic * cny + jc] = fine[IDX(i, j)]; continue; }
      const double corners = ((fine[IDX(i - 1, j - 1)] + fine[IDX(i - 1, j + 1)]) + fine[IDX(i + 1, j - 1)]) + fine[IDX(i + 1, j + 1)];
      const double edges = ((fine[IDX(i - 1, j)] + fine[IDX(i + 1, j)]) + fine[IDX(i, j - 1)]) + fine[IDX(i, j + 1)];
      coarse[(size_t)ic * cny + jc] = (1.0 / 16.0 * corners + 1.0 / 8.0 * edges) + 1.0 / 4.0 * fine[IDX(i, j)];
    }
}

/* u += P e with the reference's far-edge behaviour (fine[odd i, ny-1] = fine[nx-1, odd j] = 0) */
void mgo_prolong_add(const double* e, double* u, int nx, int ny) {
  const int cny = (ny - 1) / 2 + 1;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ny; ++j) {
      const int ic = i >> 1, jc = j >> 1, io = i & 1, jo = j & 1;
      const double* r0 = e + (size_t)ic * cny;
      const double* r1 = r0 + (io ? cny : 0);
      double v;
      if (!io && !jo) v = r0[jc];
      else if (io && !jo) v = (j == ny - 1) ? 0.0 : 0.5 * (r0[jc] + r1[jc]);
      else if (!io && jo) v = (i == nx - 1) ? 0.0 : 0.5 * (r0[jc] + r0[jc + 1]);
      else v = 0.25 * (((r0[jc] + r0[jc + 1]) + r1[jc]) + r1[jc + 1]);
      u[IDX(i, j)] += v;
    }
}

int mgo_coarse_solve(double* u, const double* f, int nx, int ny, double hx, double hy, double coeff, double tol, int maxit) {
  const double hx2 = hx * hx, hy2 = hy * hy;
  const double diag = -2.0 / hx2 - 2.0 / hy2;
  double* r = (double*)malloc(sizeof(double) * nx * ny);
  int it;
  for (it = 1; it <= maxit; ++it) {
    for (int i = 1; i < nx - 1; ++i)
      for (int j = 1; j < ny - 1; ++j) {
        const double nb = (u[IDX(i + 1, j)] + u[IDX(i - 1, j)]) / hx2 + (u[IDX(i, j + 1)] + u[IDX(i, j - 1)]) / hy2;
        const double un = (f[IDX(i, j)] + nb) / (-diag);
        u[IDX(i, j)] = (1 - 1.0) * u[IDX(i, j)] + 1.0 * un;
      }
    mgo_residual(u, f, r, nx, ny, hx, hy, coeff);
    /* NumPy's pairwise sum for < 128 elements per leaf is a plain left-to-right loop in 8 lanes; the coarsest grids
       are tiny, so the stop test may differ from NumPy's in the last ulp -- one sweep more or less, both converged */
    if (mgo_norm(r, nx, ny, hx, hy) < tol) break;
  }
  free(r);
  return it > maxit ? maxit : it;
}

typedef struct {
  int nlev;
  int nx[32], ny[32];
  double hx[32], hy[32];
  double *u[32], *t[32], *f[32], *r[32];
  double coeff, omega, ctol;
  int cycle, pre, post, smoother, cmaxit;   /* cycle: 0 V, 1 W, 2 F; smoother: 0 jacobi, 1 rbgs */
} mgo_hier;

mgo_hier* mgo_create(int nx, int ny, double x0, double x1, double y0, double y1, double coeff, int max_levels, int cycle,
                     int pre, int post, int smoother, double omega, double ctol, int cmaxit) {
  mgo_hier* h = (mgo_hier*)calloc(1, sizeof(mgo_hier));
  h->coeff = coeff; h->omega = omega; h->ctol = ctol; h->cycle = cycle; h->pre = pre; h->post = post;
  h->smoother = smoother; h->cmaxit = cmaxit;
  for (int l = 0; l < max_levels && l < 32; ++l) {
    if (l > 0) {
      if ((nx - 1) % 2 || (ny - 1) % 2) break;
      const int cx = (nx - 1) / 2 + 1, cy = (ny - 1) / 2 + 1;
      if (cx < 5 || cy < 5) break;
      nx = cx; ny = cy;
    }
    h->nx[l] = nx; h->ny[l] = ny;
    h->hx[l] = (x1 - x0) / (nx - 1); h->hy[l] = (y1 - y0) / (ny - 1);
    const size_t n = (size_t)nx * ny;
    h->u[l] = (double*)calloc(n, sizeof(double)); h->t[l] = (double*)calloc(n, sizeof(double));
    h->f[l] = (double*)calloc(n, sizeof(double)); h->r[l] = (double*)calloc(n, sizeof(double));
    h->nlev = l + 1;
  }
  return h;
}

void mgo_destroy(mgo_hier* h) {
  for (int l = 0; l < h->nlev; ++l) { free(h->u[l]); free(h->t[l]); free(h->f[l]); free(h->r[l]); }
  free(h);
}

double* mgo_u(mgo_hier* h, int l) { return h->u[l]; }
double* mgo_f(mgo_hier* h, int l) { return h->f[l]; }
int mgo_levels(mgo_hier* h) { return h->nlev; }

static void smooth(mgo_hier* h, int l, int nu) {
  const int nx = h->nx[l], ny = h->ny[l];
  for (int s = 0; s < nu; ++s) {
    if (h->smoother == 0) {
      mgo_jacobi(h->u[l], h->f[l], h->t[l], nx, ny, h->hx[l], h->hy[l], h->omega);
      double* tmp = h->u[l]; h->u[l] = h->t[l]; h->t[l] = tmp;
    } else {
      mgo_rbgs_colour(h->u[l], h->f[l], nx, ny, h->hx[l], h->hy[l], h->omega, 0);
      mgo_rbgs_colour(h->u[l], h->f[l], nx, ny, h->hx[l], h->hy[l], h->omega, 1);
    }
  }
}

void mgo_cycle(mgo_hier* h, int l) {
  const int nx = h->nx[l], ny = h->ny[l];
  if (l == h->nlev - 1) {
    mgo_coarse_solve(h->u[l], h->f[l], nx, ny, h->hx[l], h->hy[l], h->coeff, h->ctol, h->cmaxit);
    return;
  }
  smooth(h, l, h->pre);
  mgo_residual(h->u[l], h->f[l], h->r[l], nx, ny, h->hx[l], h->hy[l], h->coeff);
  mgo_restrict_fw(h->r[l], h->f[l + 1], nx, ny);
  memset(h->u[l + 1], 0, sizeof(double) * h->nx[l + 1] * h->ny[l + 1]);
  memset(h->t[l + 1], 0, sizeof(double) * h->nx[l + 1] * h->ny[l + 1]);
  int reps = 1;
  if (h->cycle == 1) reps = 2;
  else if (h->cycle == 2) { int e = h->nlev - l - 2; reps = e > 0 ? (1 << e) : 1; }
  for (int k = 0; k < reps; ++k) mgo_cycle(h, l + 1);
  mgo_prolong_add(h->u[l + 1], h->u[l], nx, ny);
  smooth(h, l, h->post);
}

double mgo_residual_norm(mgo_hier* h) {
  mgo_residual(h->u[0], h->f[0], h->r[0], h->nx[0], h->ny[0], h->hx[0], h->hy[0], h->coeff);
  return mgo_norm(h->r[0], h->nx[0], h->ny[0], h->hx[0], h->hy[0]);
}
