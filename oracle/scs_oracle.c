/*
 * oracle/scs_oracle.c -- TEST INFRASTRUCTURE ONLY, never part of the product path.
 *
 * A plain-C (C99, libm only) CPU restatement of the SCS ADMM hot path, written for
 * this repo from the reference's behaviour; every function cites the reference
 * lines (cvxgrp/scs v3.2.11, paths relative to the reference tree) it follows.
 * It is the checker that travels: /root/reference does not exist on the GPU box.
 *
 * PARITY PINNING: this restatement is checked by tests/test_oracle.py against
 *  (1) the golden vectors in tests/golden/ (captured from the real reference by
 *      tests/golden/make_golden.py): linear-system boundary pairs, cone
 *      projection pairs, whole-solve ScsInfo records;
 *  (2) the real reference itself (oracle/_ref/, built by oracle/Makefile) when
 *      that build is present.
 *
 * Scope: double precision, int32 indices, P == NULL (no BASELINE config has a
 * quadratic term), cones zero / nonneg / box / second-order / PSD, Anderson
 * acceleration off (the reference object src/aa.c stays host side and is not
 * restated here).  PSD uses a cyclic Jacobi eigensolver instead of LAPACK dsyevr.
 * Exponential and power cones (added to the HIP path after the hot path) are
 * NOT restated: the HIP kernels for them are checked directly against golden vectors
 * captured from the reference and against the reference's own exp/power test problems.
 */
#include "scs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXV(a, b) ((a) > (b) ? (a) : (b))
#define MINV(a, b) ((a) < (b) ? (a) : (b))

/* ------------------------------------------------------------------------- */
/* level-1 helpers: plain-C variants of src/linalg.c:36-102                  */
/* ------------------------------------------------------------------------- */
static double v_norm_inf(const double *a, int len) { /* linalg.c:71-82 */
  double mx = 0.0, t;
  int i;
  for (i = 0; i < len; ++i) {
    t = fabs(a[i]);
    if (t > mx) mx = t;
  }
  return mx;
}
static double v_dot(const double *x, const double *y, int len) { /* linalg.c:47-54 */
  double ip = 0.0;
  int i;
  for (i = 0; i < len; ++i) ip += x[i] * y[i];
  return ip;
}
static double safediv_pos(double x, double y) { /* glbopts.h:194-196 */
  return y < 1e-18 ? x / 1e-18 : x / y;
}

/* y += A' x for CSC A (n columns)  -- linsys/scs_matrix.c:161-186 */
void or_accum_by_atrans(int n, const int *Ap, const int *Ai, const double *Ax, const double *x, double *y) {
  int j, p;
  for (j = 0; j < n; ++j) {
    double yj = y[j];
    for (p = Ap[j]; p < Ap[j + 1]; ++p) yj += Ax[p] * x[Ai[p]];
    y[j] = yj;
  }
}
/* y += A x for CSC A  -- linsys/scs_matrix.c:188-203 */
void or_accum_by_a(int n, const int *Ap, const int *Ai, const double *Ax, const double *x, double *y) {
  int j, p;
  for (j = 0; j < n; ++j)
    for (p = Ap[j]; p < Ap[j + 1]; ++p) y[Ai[p]] += Ax[p] * x[j];
}

/* ------------------------------------------------------------------------- */
/* linear system: linsys/cpu/indirect/private.c                               */
/* ------------------------------------------------------------------------- */
struct OrLinSys {
  int n, m;
  const int *Ap, *Ai; /* borrowed CSC(A) */
  const double *Ax;
  int *Tp, *Ti; /* owned CSC(A') == CSR(A) */
  double *Tx;
  const double *diag_r; /* borrowed, like private.c:259 */
  double *p, *r, *Gp, *z, *M, *tmp;
  long tot_cg_its;
  int last_cg_its;
};

static void ls_set_preconditioner(OrLinSys *w) { /* private.c:50-82 */
  int i, k;
  for (i = 0; i < w->n; ++i) {
    double Mi = w->diag_r[i];
    for (k = w->Ap[i]; k < w->Ap[i + 1]; ++k) Mi += w->Ax[k] * w->Ax[k] / w->diag_r[w->n + w->Ai[k]];
    w->M[i] = 1. / Mi;
  }
}

OrLinSys *or_linsys_init(int m, int n, const int *Ap, const int *Ai, const double *Ax, const double *diag_r) {
  OrLinSys *w = (OrLinSys *)calloc(1, sizeof(OrLinSys));
  int nnz = Ap[n], i, j, *cnt;
  w->n = n;
  w->m = m;
  w->Ap = Ap;
  w->Ai = Ai;
  w->Ax = Ax;
  w->diag_r = diag_r;
  w->Tp = (int *)calloc(m + 1, sizeof(int));
  w->Ti = (int *)calloc(nnz > 0 ? nnz : 1, sizeof(int));
  w->Tx = (double *)calloc(nnz > 0 ? nnz : 1, sizeof(double));
  /* counting-sort transpose, private.c:7-46 */
  cnt = (int *)calloc(m, sizeof(int));
  for (i = 0; i < nnz; ++i) cnt[Ai[i]]++;
  for (i = 0; i < m; ++i) w->Tp[i + 1] = w->Tp[i] + cnt[i];
  for (i = 0; i < m; ++i) cnt[i] = w->Tp[i];
  for (j = 0; j < n; ++j)
    for (i = Ap[j]; i < Ap[j + 1]; ++i) {
      int q = cnt[Ai[i]]++;
      w->Ti[q] = j;
      w->Tx[q] = Ax[i];
    }
  free(cnt);
  w->p = (double *)calloc(n, sizeof(double));
  w->r = (double *)calloc(n, sizeof(double));
  w->Gp = (double *)calloc(n, sizeof(double));
  w->z = (double *)calloc(n, sizeof(double));
  w->M = (double *)calloc(n, sizeof(double));
  w->tmp = (double *)calloc(m, sizeof(double));
  ls_set_preconditioner(w);
  return w;
}

void or_linsys_update_diag_r(OrLinSys *w, const double *diag_r) { /* private.c:327-331 */
  w->diag_r = diag_r;
  ls_set_preconditioner(w);
}

void or_linsys_free(OrLinSys *w) {
  if (!w) return;
  free(w->Tp); free(w->Ti); free(w->Tx);
  free(w->p); free(w->r); free(w->Gp); free(w->z); free(w->M); free(w->tmp);
  free(w);
}

/* y = (R_x + A' R_y^-1 A) x   -- private.c:106-119 */
static void ls_mat_vec(OrLinSys *w, const double *x, double *y) {
  int i;
  memset(w->tmp, 0, w->m * sizeof(double));
  memset(y, 0, w->n * sizeof(double));
  or_accum_by_atrans(w->m, w->Tp, w->Ti, w->Tx, x, w->tmp); /* tmp = A x (via the stored transpose) */
  for (i = 0; i < w->m; ++i) w->tmp[i] /= w->diag_r[w->n + i];
  or_accum_by_atrans(w->n, w->Ap, w->Ai, w->Ax, w->tmp, y);
  for (i = 0; i < w->n; ++i) y[i] += w->diag_r[i] * x[i];
}

/* private.c:133-217 */
static int ls_pcg(OrLinSys *w, const double *s, double *b, int max_its, double tol) {
  int i, k, n = w->n;
  double ztr, ztr_prev, alpha;
  double *p = w->p, *Gp = w->Gp, *r = w->r, *z = w->z, *M = w->M;
  if (!s) {
    memcpy(r, b, n * sizeof(double));
    memset(b, 0, n * sizeof(double));
  } else {
    ls_mat_vec(w, s, r);
    for (k = 0; k < n; ++k) r[k] += -1. * b[k];
    for (k = 0; k < n; ++k) r[k] *= -1.;
    memcpy(b, s, n * sizeof(double));
  }
  if (v_norm_inf(r, n) < MAXV(tol, 1e-12)) return 0;
  for (k = 0; k < n; ++k) z[k] = r[k] * M[k];
  ztr = v_dot(z, r, n);
  memcpy(p, z, n * sizeof(double));
  for (i = 0; i < max_its; ++i) {
    double norm_r = 0.0, beta;
    ls_mat_vec(w, p, Gp);
    alpha = ztr / v_dot(p, Gp, n);
    for (k = 0; k < n; ++k) b[k] += alpha * p[k];
    for (k = 0; k < n; ++k) r[k] += -alpha * Gp[k];
    ztr_prev = ztr;
    ztr = 0.0;
    for (k = 0; k < n; ++k) {
      double rk = r[k], zk = rk * M[k], ark = fabs(rk);
      z[k] = zk;
      ztr += zk * rk;
      if (ark > norm_r) norm_r = ark;
    }
    if (norm_r < tol) return i + 1;
    if (ztr_prev == 0.) break;
    beta = ztr / ztr_prev;
    for (k = 0; k < n; ++k) p[k] = z[k] + beta * p[k];
  }
  return i;
}

/* private.c:284-324 */
int or_linsys_solve(OrLinSys *w, double *b, const double *s, double tol) {
  int i, n = w->n, m = w->m, its;
  if (v_norm_inf(b, n + m) <= 1e-12) {
    memset(b, 0, (n + m) * sizeof(double));
    w->last_cg_its = 0;
    return 0;
  }
  memcpy(w->tmp, b + n, m * sizeof(double));
  for (i = 0; i < m; ++i) w->tmp[i] /= w->diag_r[n + i];
  or_accum_by_atrans(n, w->Ap, w->Ai, w->Ax, w->tmp, b);
  its = ls_pcg(w, s, b, 10 * n, tol);
  for (i = 0; i < m; ++i) b[n + i] *= -1.;
  or_accum_by_atrans(m, w->Tp, w->Ti, w->Tx, b, b + n);
  for (i = 0; i < m; ++i) b[n + i] /= w->diag_r[n + i];
  w->tot_cg_its += its;
  w->last_cg_its = its;
  return 0;
}
long or_linsys_tot_cg_its(const OrLinSys *w) { return w->tot_cg_its; }

/* ------------------------------------------------------------------------- */
/* cones: src/cones.c                                                         */
/* ------------------------------------------------------------------------- */
static void cone_soc(double *x, int q) { /* cones.c:1250-1279 */
  double v1, s, alpha;
  int i;
  if (q <= 0) return;
  if (q == 1) {
    x[0] = MAXV(x[0], 0.);
    return;
  }
  v1 = x[0];
  if (q == 2) s = fabs(x[1]);
  else if (q == 3) s = sqrt(x[1] * x[1] + x[2] * x[2]);
  else {
    s = 0;
    for (i = 1; i < q; ++i) s += x[i] * x[i];
    s = sqrt(s);
  }
  alpha = (s + v1) / 2.0;
  if (s <= v1) return;
  if (s <= -v1) memset(x, 0, q * sizeof(double));
  else {
    x[0] = alpha;
    for (i = 1; i < q; ++i) x[i] *= alpha / s;
  }
}

/* cones.c:1182-1245 */
static double cone_box(double *tx, const double *bl, const double *bu, int bsize, double t_wm, const double *r_box) {
  double *x = tx + 1, gt, ht, t = t_wm, t_prev, r, rho_t = 1.0;
  const double *rho = NULL;
  int iter, j;
  if (bsize == 1) {
    tx[0] = MAXV(tx[0], 0.0);
    return tx[0];
  }
  if (r_box) {
    rho_t = 1.0 / r_box[0];
    rho = r_box + 1;
  }
  for (iter = 0; iter < 25; iter++) {
    t_prev = t;
    gt = rho_t * (t - tx[0]);
    ht = rho_t;
    for (j = 0; j < bsize - 1; j++) {
      r = rho ? 1.0 / rho[j] : 1.0;
      if (x[j] > t * bu[j]) {
        gt += r * (t * bu[j] - x[j]) * bu[j];
        ht += r * bu[j] * bu[j];
      } else if (x[j] < t * bl[j]) {
        gt += r * (t * bl[j] - x[j]) * bl[j];
        ht += r * bl[j] * bl[j];
      }
    }
    t = MAXV(t - gt / MAXV(ht, 1e-8), 0.0);
    if (fabs(gt / MAXV(ht, 1e-6)) < 1e-12 * MAXV(t, 1.) || fabs(t - t_prev) < 1e-11 * MAXV(t, 1.)) break;
  }
  for (j = 0; j < bsize - 1; j++) {
    if (x[j] > t * bu[j]) x[j] = t * bu[j];
    else if (x[j] < t * bl[j]) x[j] = t * bl[j];
  }
  tx[0] = t;
  return t;
}

/* cones.c:999-1067 with the LAPACK dsyevr/dsyrk pair replaced by cyclic Jacobi */
static void cone_psd(double *X, int k) {
  double *A, *V, sqrt2 = sqrt(2.0), fro = 0;
  int i, j, c, sweep, p, q;
  if (k == 0) return;
  if (k == 1) {
    X[0] = MAXV(X[0], 0.);
    return;
  }
  A = (double *)calloc((size_t)k * k, sizeof(double));
  V = (double *)calloc((size_t)k * k, sizeof(double));
  for (j = 0; j < k; ++j)
    for (i = j; i < k; ++i) { /* packed lower triangle, column major (cones.c:1018-1021) */
      double v = X[j * k - (j * (j - 1)) / 2 + (i - j)];
      if (i == j) v *= sqrt2;
      A[i * k + j] = A[j * k + i] = v;
    }
  for (i = 0; i < k; ++i) V[i * k + i] = 1.0;
  for (i = 0; i < k * k; ++i) fro += A[i] * A[i];
  fro = sqrt(fro);
  for (sweep = 0; sweep < 60 && fro > 0; ++sweep) {
    double off = 0;
    for (p = 0; p < k - 1; ++p)
      for (q = p + 1; q < k; ++q) {
        double apq = A[p * k + q], theta, t, cs, sn;
        if (fabs(apq) > off) off = fabs(apq);
        if (apq == 0.0) continue;
        theta = (A[q * k + q] - A[p * k + p]) / (2.0 * apq);
        t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        cs = 1.0 / sqrt(t * t + 1.0);
        sn = t * cs;
        for (j = 0; j < k; ++j) { /* rows p, q */
          double ap = A[p * k + j], aq = A[q * k + j];
          A[p * k + j] = cs * ap - sn * aq;
          A[q * k + j] = sn * ap + cs * aq;
        }
        for (i = 0; i < k; ++i) { /* columns p, q of A and V */
          double ap = A[i * k + p], aq = A[i * k + q], vp = V[i * k + p], vq = V[i * k + q];
          A[i * k + p] = cs * ap - sn * aq;
          A[i * k + q] = sn * ap + cs * aq;
          V[i * k + p] = cs * vp - sn * vq;
          V[i * k + q] = sn * vp + cs * vq;
        }
      }
    if (off <= 1e-16 * fro) break;
  }
  /* X+ = sum_{lambda > 0} lambda v v'  (cones.c:1036-1052); diagonal / sqrt(2) (:1055) */
  for (j = 0; j < k; ++j)
    for (i = j; i < k; ++i) {
      double acc = 0;
      for (c = 0; c < k; ++c) {
        double lam = A[c * k + c];
        if (lam > 0) acc += lam * V[i * k + c] * V[j * k + c];
      }
      if (i == j) acc /= sqrt2;
      X[j * k - (j * (j - 1)) / 2 + (i - j)] = acc;
    }
  free(A);
  free(V);
}

struct OrCone {
  int m, z, l, bsize, qsize, ssize;
  int *q, *s;
  double *bl, *bu; /* private, normalised copies */
  double box_t_warm_start;
  double *scratch;
};

/* init_cone (cones.c:1498-1538) + the lazy normalize_box_cone (:1161-1177, :1557-1565) */
OrCone *or_cone_init(int m, int z, int l, int bsize, const double *bl, const double *bu, int qsize, const int *q,
                     int ssize, const int *s, const double *D) {
  OrCone *c = (OrCone *)calloc(1, sizeof(OrCone));
  int j;
  c->m = m; c->z = z; c->l = l; c->bsize = bsize; c->qsize = qsize; c->ssize = ssize;
  c->q = (int *)calloc(qsize > 0 ? qsize : 1, sizeof(int));
  c->s = (int *)calloc(ssize > 0 ? ssize : 1, sizeof(int));
  if (qsize) memcpy(c->q, q, qsize * sizeof(int));
  if (ssize) memcpy(c->s, s, ssize * sizeof(int));
  c->bl = (double *)calloc(bsize > 1 ? bsize - 1 : 1, sizeof(double));
  c->bu = (double *)calloc(bsize > 1 ? bsize - 1 : 1, sizeof(double));
  for (j = 0; j < bsize - 1; ++j) {
    const double *Db = D ? D + z + l : NULL;
    double f = Db ? Db[j + 1] / Db[0] : 1.0;
    c->bu[j] = bu[j] >= 1e15 ? INFINITY : bu[j] * f;
    c->bl[j] = bl[j] <= -1e15 ? -INFINITY : bl[j] * f;
  }
  c->box_t_warm_start = 1.;
  c->scratch = (double *)calloc(m > 0 ? m : 1, sizeof(double));
  return c;
}
void or_cone_free(OrCone *c) {
  if (!c) return;
  free(c->q); free(c->s); free(c->bl); free(c->bu); free(c->scratch);
  free(c);
}

/* proj_cone, cones.c:1340-1394 (zero, nonneg, box, SOC, PSD in that order) */
static void cone_proj_primal(OrCone *c, double *x, const double *r_y) {
  int i, count = 0;
  if (c->z) {
    memset(x, 0, c->z * sizeof(double));
    count += c->z;
  }
  for (i = count; i < count + c->l; ++i) x[i] = MAXV(x[i], 0.0);
  count += c->l;
  if (c->bsize) {
    c->box_t_warm_start = cone_box(x + count, c->bl, c->bu, c->bsize, c->box_t_warm_start, r_y ? r_y + count : NULL);
    count += c->bsize;
  }
  for (i = 0; i < c->qsize; ++i) {
    cone_soc(x + count, c->q[i]);
    count += c->q[i];
  }
  for (i = 0; i < c->ssize; ++i) {
    cone_psd(x + count, c->s[i]);
    count += c->s[i] * (c->s[i] + 1) / 2;
  }
}

/* proj_dual_cone, cones.c:1552-1596 */
void or_cone_proj_dual(OrCone *c, double *x, const double *r_y) {
  int i;
  memcpy(c->scratch, x, c->m * sizeof(double));
  if (r_y) for (i = 0; i < c->m; ++i) x[i] *= -r_y[i];
  else for (i = 0; i < c->m; ++i) x[i] = -x[i];
  cone_proj_primal(c, x, r_y);
  if (r_y) for (i = 0; i < c->m; ++i) x[i] = x[i] / r_y[i] + c->scratch[i];
  else for (i = 0; i < c->m; ++i) x[i] += c->scratch[i];
}

/* ------------------------------------------------------------------------- */
/* equilibration: linsys/scs_matrix.c:229-496, src/normalize.c:33-91          */
/* ------------------------------------------------------------------------- */
static double apply_limit(double x) {
  x = x < 1e-4 ? 1.0 : x;
  x = x > 1e4 ? 1e4 : x;
  return x;
}
/* enforce_cone_boundaries (cones.c:366-379): how = 0 max |.|, 1 mean */
static void enforce_boundaries(const OrCone *c, double *vec, int how) {
  int count = c->z + c->l + c->bsize, i, j, nc = c->qsize + c->ssize;
  for (i = 0; i < nc; ++i) {
    int delta = i < c->qsize ? c->q[i] : c->s[i - c->qsize] * (c->s[i - c->qsize] + 1) / 2;
    double w = 0;
    if (how == 0) w = v_norm_inf(vec + count, delta);
    else if (delta > 0) {
      for (j = 0; j < delta; ++j) w += vec[count + j];
      w /= delta;
    }
    for (j = count; j < count + delta; ++j) vec[j] = w;
    count += delta;
  }
}
static void or_normalize_a(int m, int n, const int *Ap, const int *Ai, double *Ax, const OrCone *cone, double *D,
                           double *E) {
  double *Dt = (double *)calloc(m, sizeof(double)), *Et = (double *)calloc(n, sizeof(double));
  int pass, i, j;
  for (i = 0; i < m; ++i) D[i] = 1.;
  for (i = 0; i < n; ++i) E[i] = 1.;
  for (pass = 0; pass < 26; ++pass) {
    const int l2 = pass == 25; /* 25 Ruiz passes then one L2 pass (scs_matrix.c:15-16,470-477) */
    for (i = 0; i < m; ++i) Dt[i] = 0.;
    for (i = 0; i < n; ++i)
      for (j = Ap[i]; j < Ap[i + 1]; ++j) {
        if (l2) Dt[Ai[j]] += Ax[j] * Ax[j];
        else Dt[Ai[j]] = MAXV(Dt[Ai[j]], fabs(Ax[j]));
      }
    if (l2) for (i = 0; i < m; ++i) Dt[i] = sqrt(Dt[i]);
    enforce_boundaries(cone, Dt, l2);
    for (i = 0; i < m; ++i) Dt[i] = safediv_pos(1.0, sqrt(apply_limit(Dt[i])));
    for (i = 0; i < n; ++i) {
      double e = 0, t;
      for (j = Ap[i]; j < Ap[i + 1]; ++j) {
        if (l2) e += Ax[j] * Ax[j];
        else {
          t = fabs(Ax[j]);
          if (t > e) e = t;
        }
      }
      if (l2) e = sqrt(e);
      Et[i] = safediv_pos(1.0, sqrt(apply_limit(e)));
    }
    for (i = 0; i < n; ++i) /* rescale, scs_matrix.c:370-407 */
      for (j = Ap[i]; j < Ap[i + 1]; ++j) Ax[j] *= Dt[Ai[j]] * Et[i];
    for (i = 0; i < m; ++i) D[i] *= Dt[i];
    for (i = 0; i < n; ++i) E[i] *= Et[i];
  }
  free(Dt);
  free(Et);
}

/* ------------------------------------------------------------------------- */
/* the ADMM loop: src/scs.c                                                   */
/* ------------------------------------------------------------------------- */
typedef struct {
  double tau, kap, bty_tau, ctx_tau, bty, ctx, gap, pobj, dobj;
  double res_pri, res_dual, res_infeas, res_unbdd_a, res_unbdd_p;
  double nm_pri, nm_dual, nm_ax_s, nm_ax, nm_aty, nm_s;
} OrResid;

static double root_plus_coeffs(double a, double b, double c) { /* scs.c:689-708 */
  double rad, sq, q;
  if (!isfinite(a) || !isfinite(b) || !isfinite(c) || a <= 0.) return NAN;
  rad = b * b - 4 * a * c;
  if (!isfinite(rad)) return NAN;
  if (rad < 0.) return -b / (2 * a);
  sq = sqrt(rad);
  if (b <= 0.) return (-b + sq) / (2 * a);
  q = -0.5 * (b + sq);
  return q != 0. ? c / q : 0.;
}

static void resid_finish(OrResid *r, double pd) { /* compute_residuals, scs.c:463-485 */
  double tol = 1e-9 / pd;
  r->res_pri = safediv_pos(r->nm_pri, r->tau);
  r->res_dual = safediv_pos(r->nm_dual, r->tau);
  r->res_unbdd_a = r->res_unbdd_p = r->res_infeas = NAN;
  if (r->ctx_tau < -tol) {
    r->res_unbdd_a = safediv_pos(r->nm_ax_s, -r->ctx_tau);
    r->res_unbdd_p = safediv_pos(0.0, -r->ctx_tau);
  }
  if (r->bty_tau < -tol) r->res_infeas = safediv_pos(r->nm_aty, -r->bty_tau);
}

/* populate_residual_struct (scs.c:535-607) + unnormalize_residuals (:487-531) */
static void or_residuals(int m, int n, const int *Ap, const int *Ai, const double *Ax, const double *b,
                         const double *c, const double *D, const double *E, double sigma, const double *u,
                         const double *rsk, double *ax, double *aty, OrResid *rn_, OrResid *ro_) {
  const int L = n + m + 1;
  int i;
  OrResid rn, ro;
  const double *x = u, *y = u + n, *sv = rsk + n;
  double pd = sigma * sigma;
  memset(&rn, 0, sizeof rn);
  memset(&ro, 0, sizeof ro);
  rn.tau = fabs(u[L - 1]);
  rn.kap = fabs(rsk[L - 1]);
  memset(ax, 0, m * sizeof(double));
  or_accum_by_a(n, Ap, Ai, Ax, x, ax);
  memset(aty, 0, n * sizeof(double));
  or_accum_by_atrans(n, Ap, Ai, Ax, y, aty);
  for (i = 0; i < m; ++i) {
    double axs = ax[i] + sv[i], pri = axs - rn.tau * b[i], f = (1.0 / sigma) / D[i];
    rn.nm_pri = MAXV(rn.nm_pri, fabs(pri)); rn.nm_ax_s = MAXV(rn.nm_ax_s, fabs(axs));
    rn.nm_ax = MAXV(rn.nm_ax, fabs(ax[i])); rn.nm_s = MAXV(rn.nm_s, fabs(sv[i]));
    ro.nm_pri = MAXV(ro.nm_pri, fabs(pri * f)); ro.nm_ax_s = MAXV(ro.nm_ax_s, fabs(axs * f));
    ro.nm_ax = MAXV(ro.nm_ax, fabs(ax[i] * f)); ro.nm_s = MAXV(ro.nm_s, fabs(sv[i] / (D[i] * sigma)));
  }
  for (i = 0; i < n; ++i) {
    double du = aty[i] + rn.tau * c[i], f = (1.0 / sigma) / E[i];
    rn.nm_dual = MAXV(rn.nm_dual, fabs(du)); rn.nm_aty = MAXV(rn.nm_aty, fabs(aty[i]));
    ro.nm_dual = MAXV(ro.nm_dual, fabs(du * f)); ro.nm_aty = MAXV(ro.nm_aty, fabs(aty[i] * f));
  }
  rn.bty_tau = v_dot(y, b, m);
  rn.ctx_tau = v_dot(x, c, n);
  rn.bty = safediv_pos(rn.bty_tau, rn.tau);
  rn.ctx = safediv_pos(rn.ctx_tau, rn.tau);
  rn.gap = fabs(rn.ctx + rn.bty);
  rn.pobj = rn.ctx;
  rn.dobj = -rn.bty;
  resid_finish(&rn, 1.0);
  ro.tau = rn.tau; ro.kap = rn.kap / pd; ro.bty_tau = rn.bty_tau / pd; ro.ctx_tau = rn.ctx_tau / pd;
  ro.bty = rn.bty / pd; ro.ctx = rn.ctx / pd; ro.gap = rn.gap / pd; ro.pobj = rn.pobj / pd; ro.dobj = rn.dobj / pd;
  resid_finish(&ro, pd);
  *rn_ = rn;
  *ro_ = ro;
}

int or_solve(int m, int n, const int *Ap, const int *Ai, const double *Ax_in, const double *b_in, const double *c_in,
             int z, int l, int bsize, const double *bl, const double *bu, int qsize, const int *q, int ssize,
             const int *s, const OrSettings *st, double *xo, double *yo, double *so, OrInfo *info) {
  const int L = n + m + 1, nnz = Ap[n];
  int i, iter, status = 0, last_resid_iter = -1, last_scale_update_iter = 0, n_log = 0, scale_updates = 0;
  double sum_log = 0, scale = st->scale, sigma = 1.0, nm_b_orig, nm_c_orig;
  double *Ax = (double *)malloc((nnz > 0 ? nnz : 1) * sizeof(double));
  double *b = (double *)malloc(m * sizeof(double)), *c = (double *)malloc(n * sizeof(double));
  double *D = (double *)malloc(m * sizeof(double)), *E = (double *)malloc(n * sizeof(double));
  double *u = (double *)calloc(L, sizeof(double)), *u_t = (double *)calloc(L, sizeof(double));
  double *v = (double *)calloc(L, sizeof(double)), *rsk = (double *)calloc(L, sizeof(double));
  double *g = (double *)calloc(L, sizeof(double)), *R = (double *)calloc(L, sizeof(double));
  double *warm = (double *)calloc(n, sizeof(double));
  double *ax = (double *)calloc(m, sizeof(double)), *aty = (double *)calloc(n, sizeof(double));
  OrResid rn, ro;
  OrCone *cone;
  OrLinSys *ls;
  memset(&rn, 0, sizeof rn);
  memset(&ro, 0, sizeof ro);
  memcpy(Ax, Ax_in, nnz * sizeof(double));
  memcpy(b, b_in, m * sizeof(double));
  memcpy(c, c_in, n * sizeof(double));
  nm_b_orig = v_norm_inf(b, m);
  nm_c_orig = v_norm_inf(c, n);
  for (i = 0; i < m; ++i) D[i] = 1.;
  for (i = 0; i < n; ++i) E[i] = 1.;
  cone = or_cone_init(m, z, l, bsize, bl, bu, qsize, q, ssize, s, NULL);
  if (st->normalize) {
    double nb, nc;
    or_normalize_a(m, n, Ap, Ai, Ax, cone, D, E);
    for (i = 0; i < n; ++i) c[i] *= E[i]; /* normalize_b_c, normalize.c:33-61 */
    for (i = 0; i < m; ++i) b[i] *= D[i];
    nc = v_norm_inf(c, n);
    nb = v_norm_inf(b, m);
    sigma = MAXV(nc, nb);
    sigma = sigma < 1e-4 ? 1.0 : sigma;
    sigma = sigma > 1e4 ? 1e4 : sigma;
    sigma = safediv_pos(1.0, sigma);
    for (i = 0; i < n; ++i) c[i] *= sigma;
    for (i = 0; i < m; ++i) b[i] *= sigma;
    or_cone_free(cone); /* box bounds pick up D on first projection (cones.c:1557-1565) */
    cone = or_cone_init(m, z, l, bsize, bl, bu, qsize, q, ssize, s, D);
  }
#define SET_DIAG_R()                                                                               \
  do { /* scs.c:971-980 + cones.c:349-363 */                                                       \
    for (i = 0; i < n; ++i) R[i] = st->rho_x;                                                      \
    for (i = 0; i < z; ++i) R[n + i] = 1.0 / (1000. * scale);                                      \
    for (i = z; i < m; ++i) R[n + i] = 1.0 / scale;                                                \
    R[n + m] = 10.;                                                                                \
  } while (0)
#define UPDATE_WORK_CACHE()                                                                        \
  do { /* scs.c:1118-1128 */                                                                       \
    memcpy(g, c, n * sizeof(double));                                                              \
    for (i = 0; i < m; ++i) g[n + i] = -b[i];                                                      \
    or_linsys_solve(ls, g, NULL, 1e-12);                                                           \
  } while (0)
  SET_DIAG_R();
  ls = or_linsys_init(m, n, Ap, Ai, Ax, R);
  v[L - 1] = 1.; /* cold start, scs.c:681-685 */
  UPDATE_WORK_CACHE();

  for (iter = 0; iter < st->max_iters; ++iter) {
    double tol, nm_ws, tau_t;
    if (iter >= 1) { /* normalize_v, scs.c:813-821 */
      double nv = sqrt(v_dot(v, v, L));
      if (nv != 0.) {
        double f = sqrt((double)L) * 1. / nv;
        for (i = 0; i < L; ++i) v[i] *= f;
      }
    }
    /* project_lin_sys, scs.c:733-771 */
    for (i = 0; i < n; ++i) u_t[i] = v[i] * R[i];
    for (i = n; i < L - 1; ++i) u_t[i] = -v[i] * R[i];
    u_t[L - 1] = v[L - 1];
    memcpy(warm, u, n * sizeof(double));
    for (i = 0; i < n; ++i) warm[i] += u[L - 1] * g[i];
    tol = MINV(rn.nm_pri, rn.nm_dual);
    nm_ws = v_norm_inf(warm, n) / pow((double)iter + 1, 1.5);
    tol = 0.2 * MINV(tol, nm_ws);
    tol = MAXV(1e-12, tol);
    if (st->cg_tol_override > 0) tol = st->cg_tol_override;
    or_linsys_solve(ls, u_t, warm, tol);
    if (iter < 1) tau_t = 1.;
    else { /* root_plus, scs.c:710-730 */
      double gg = 0, mug = 0, pg = 0, pp = 0, pmu = 0, ts = R[L - 1];
      for (i = 0; i < L - 1; ++i) {
        double ri = R[i], gi = g[i], pi = u_t[i], mui = v[i];
        gg += gi * gi * ri; mug += mui * gi * ri; pg += pi * gi * ri; pp += pi * pi * ri; pmu += pi * mui * ri;
      }
      tau_t = root_plus_coeffs(ts + gg, mug - 2 * pg - v[L - 1] * ts, pp - pmu);
    }
    u_t[L - 1] = tau_t;
    for (i = 0; i < L - 1; ++i) u_t[i] += -tau_t * g[i];
    /* project_cones, scs.c:796-810 */
    for (i = 0; i < L; ++i) u[i] = 2 * u_t[i] - v[i];
    or_cone_proj_dual(cone, u + n, R + n);
    u[L - 1] = iter < 1 ? 1.0 : MAXV(u[L - 1], 0.);
    /* compute_rsk, scs.c:781-786 */
    for (i = 0; i < L; ++i) rsk[i] = (v[i] + u[i] - 2 * u_t[i]) * R[i];

    if (iter % 25 == 0) {
      last_resid_iter = iter;
      or_residuals(m, n, Ap, Ai, Ax, b, c, D, E, sigma, u, rsk, ax, aty, &rn, &ro);
      { /* has_converged, scs.c:611-649 */
        if (ro.tau > 0.) {
          double grl = MAXV(fabs(ro.ctx), fabs(ro.bty));
          double prl = MAXV(MAXV(nm_b_orig * ro.tau, ro.nm_s), ro.nm_ax) / ro.tau;
          double drl = MAXV(nm_c_orig * ro.tau, ro.nm_aty) / ro.tau;
          if (isless(ro.res_pri, st->eps_abs + st->eps_rel * prl) &&
              isless(ro.res_dual, st->eps_abs + st->eps_rel * drl) && isless(ro.gap, st->eps_abs + st->eps_rel * grl))
            status = 1;
        }
        if (!status && isless(ro.res_unbdd_a, st->eps_infeas) && isless(ro.res_unbdd_p, st->eps_infeas)) status = -1;
        if (!status && isless(ro.res_infeas, st->eps_infeas)) status = -2;
        if (status) break;
      }
    }
    if (st->adaptive_scale && iter == last_resid_iter) { /* update_scale, scs.c:1164-1241 */
      double rp, rd, factor, new_scale;
      rp = safediv_pos(ro.nm_pri, MAXV(MAXV(ro.nm_ax, ro.nm_s), nm_b_orig * ro.tau));
      rd = safediv_pos(ro.nm_dual, MAXV(ro.nm_aty, nm_c_orig * ro.tau));
      rp = MAXV(rp, 1e-18);
      rd = MAXV(rd, 1e-18);
      sum_log += log(rp) - log(rd);
      n_log++;
      factor = sqrt(exp(sum_log / (double)n_log));
      if (iter - last_scale_update_iter >= 100) {
        new_scale = MINV(MAXV(scale * factor, 1e-6), 1e6);
        if (new_scale != scale && (factor > sqrt(10.) || factor < 1. / sqrt(10.))) {
          scale_updates++;
          sum_log = 0;
          n_log = 0;
          last_scale_update_iter = iter;
          scale = new_scale;
          SET_DIAG_R();
          or_linsys_update_diag_r(ls, R);
          UPDATE_WORK_CACHE();
          for (i = 0; i < L; ++i) v[i] = rsk[i] / R[i] + 2 * u_t[i] - u[i];
        }
      }
    }
    for (i = 0; i < L; ++i) v[i] += st->alpha * (u[i] - u_t[i]); /* update_dual_vars, scs.c:788-793 */
  }
  /* finalize (scs.c:916-969) for the solved / unfinished-with-tau>0 cases */
  if (last_resid_iter != iter) or_residuals(m, n, Ap, Ai, Ax, b, c, D, E, sigma, u, rsk, ax, aty, &rn, &ro);
  for (i = 0; i < n; ++i) xo[i] = u[i] * (E[i] / sigma);
  for (i = 0; i < m; ++i) {
    yo[i] = u[n + i] * (D[i] / sigma);
    so[i] = rsk[n + i] / (D[i] * sigma);
  }
  {
    double it = safediv_pos(1.0, ro.tau);
    for (i = 0; i < n; ++i) xo[i] *= it;
    for (i = 0; i < m; ++i) {
      yo[i] *= it;
      so[i] *= it;
    }
  }
  info->iter = iter;
  info->status_val = status ? status : 2;
  info->scale_updates = scale_updates;
  info->pobj = ro.ctx;
  info->dobj = -ro.bty;
  info->res_pri = ro.res_pri;
  info->res_dual = ro.res_dual;
  info->gap = ro.gap;
  info->scale = scale;
  info->cg_its = (double)or_linsys_tot_cg_its(ls);
  or_linsys_free(ls);
  or_cone_free(cone);
  free(Ax); free(b); free(c); free(D); free(E); free(u); free(u_t); free(v); free(rsk); free(g); free(R);
  free(warm); free(ax); free(aty);
  return info->status_val;
}
