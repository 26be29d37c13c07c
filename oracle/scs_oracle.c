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
 * quadratic term), Anderson acceleration off (src/aa.c is restated separately, in the
 * product's host AA, and pinned against the reference object in tests/test_aa_host.py).
 * Cone PROJECTIONS: every cone the HIP path carries -- zero / nonneg / box / second-order /
 * PSD / complex PSD / exponential (primal, dual) / power (primal, dual); PSD and complex
 * PSD use a cyclic Jacobi eigensolver instead of LAPACK dsyevr / zheevr.  The whole-solve
 * restatement takes the same set (or_solve: the BASELINE cone families; or_solve_ext: all of
 * them) and walks the reference's trajectory on each (tests/test_oracle.py).
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

/* PSD part of a full symmetric k x k matrix (row major, overwritten): cyclic Jacobi in place of the
 * LAPACK dsyevr/dsyrk pair of cones.c:1028-1052 */
static void psd_part_full(double *A, int k) {
  double *V, fro = 0;
  int i, j, c, sweep, p, q;
  V = (double *)calloc((size_t)k * k, sizeof(double));
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
  { /* A <- sum_{lambda > 0} lambda v v' */
    double *lam = (double *)calloc((size_t)k, sizeof(double));
    for (c = 0; c < k; ++c) lam[c] = A[c * k + c];
    for (i = 0; i < k; ++i)
      for (j = 0; j < k; ++j) {
        double acc = 0;
        for (c = 0; c < k; ++c)
          if (lam[c] > 0) acc += lam[c] * V[i * k + c] * V[j * k + c];
        A[i * k + j] = acc;
      }
    free(lam);
  }
  free(V);
}

/* cones.c:999-1067: packed lower triangle (column major, off-diagonals pre-scaled by sqrt 2) */
static void cone_psd(double *X, int k) {
  double *A, sqrt2 = sqrt(2.0);
  int i, j;
  if (k == 0) return;
  if (k == 1) {
    X[0] = MAXV(X[0], 0.);
    return;
  }
  A = (double *)calloc((size_t)k * k, sizeof(double));
  for (j = 0; j < k; ++j)
    for (i = j; i < k; ++i) { /* cones.c:1018-1025 */
      double v = X[j * k - (j * (j - 1)) / 2 + (i - j)];
      if (i == j) v *= sqrt2;
      A[i * k + j] = A[j * k + i] = v;
    }
  psd_part_full(A, k);
  for (j = 0; j < k; ++j)
    for (i = j; i < k; ++i) { /* diagonal / sqrt(2), cones.c:1055 */
      double v = A[i * k + j];
      if (i == j) v /= sqrt2;
      X[j * k - (j * (j - 1)) / 2 + (i - j)] = v;
    }
  free(A);
}

/* complex Hermitian PSD cone, cones.c:1072-1155 (LAPACK zheevr there).  Packed layout: column c
 * of the lower triangle starts at c (2 nn - c): the real diagonal entry, then (re, im) pairs of
 * rows c+1..nn-1, off-diagonals pre-scaled by sqrt 2.  Solved through the real symmetric embedding
 * M = [[A, -B], [B, A]] of H = A + iB: the PSD part of M is the embedding of the PSD part of H. */
static void cone_cpsd(double *X, int nn) {
  const int K = 2 * nn;
  double *M, sqrt2 = sqrt(2.0);
  int r, c;
  if (nn == 0) return;
  if (nn == 1) {
    X[0] = MAXV(X[0], 0.);
    return;
  }
  M = (double *)calloc((size_t)K * K, sizeof(double));
  for (c = 0; c < nn; ++c) {
    const double *col = X + c * (2 * nn - c);
    const double d = col[0] * sqrt2;
    M[c * K + c] = d;
    M[(c + nn) * K + (c + nn)] = d;
    for (r = c + 1; r < nn; ++r) {
      const double re = col[1 + 2 * (r - c - 1)], im = col[2 + 2 * (r - c - 1)]; /* H[r][c] */
      M[r * K + c] = M[c * K + r] = re;                         /* A */
      M[(r + nn) * K + (c + nn)] = M[(c + nn) * K + (r + nn)] = re;
      M[(r + nn) * K + c] = M[c * K + (r + nn)] = im;           /* B[r][c] =  im */
      M[(c + nn) * K + r] = M[r * K + (c + nn)] = -im;          /* B[c][r] = -im */
    }
  }
  psd_part_full(M, K);
  for (c = 0; c < nn; ++c) {
    double *col = X + c * (2 * nn - c);
    col[0] = M[c * K + c] / sqrt2;
    for (r = c + 1; r < nn; ++r) {
      col[1 + 2 * (r - c - 1)] = M[r * K + c];
      col[2 + 2 * (r - c - 1)] = M[(r + nn) * K + c];
    }
  }
  free(M);
}

/* ---- exponential cone: src/exp_cone.c:373-441 `proj_pd_exp_cone` (Friberg 2021: heuristic
 * primal / polar points, optimality shortcut, bracket, damped Newton + bisection on h(rho)) ---- */
#define EXP_INF 1e15
static int ex_finite(double x) { return fabs(x) < EXP_INF; }
static double ex_clip(double x, double l, double u) { return MAXV(l, MINV(u, x)); }
static double ex_safediv(double x, double y) { return y < 1e-18 ? x / 1e-18 : x / y; }
static double ex_dist_sq(const double *a, const double *b) {
  const double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  return d0 * d0 + d1 * d1 + d2 * d2;
}
static void ex_h(const double *v0, double rho, double *f, double *df) { /* exp_cone.c:41-64 */
  const double t0 = v0[2], s0 = v0[1], r0 = v0[0];
  const double er = exp(rho), enr = 1.0 / er;
  *f = ((rho - 1) * r0 + s0) * er - (r0 - rho * s0) * enr - (rho * (rho - 1) + 1) * t0;
  if (df) *df = (rho * r0 + s0) * er + (r0 - (rho - 1) * s0) * enr - (2 * rho - 1) * t0;
}
static double ex_bisect(const double *v0, double xl, double xu, double x) { /* :67-98 */
  double xp = x, f;
  int i;
  for (i = 0; i < 40; ++i) {
    ex_h(v0, x, &f, NULL);
    if (f < 0.0) xl = x;
    else xu = x;
    xp = 0.5 * (xl + xu);
    if (fabs(xp - x) <= 1e-12 * MAXV(1.0, fabs(xp)) || xp == xl || xp == xu) break;
    x = xp;
  }
  return xp;
}
static double ex_newton(const double *v0, double xl, double xu, double x) { /* :101-157 */
  double xp, f, df;
  int i;
  for (i = 0; i < 20; ++i) {
    ex_h(v0, x, &f, &df);
    if (fabs(f) <= 1e-15) break;
    if (f < 0.0) xl = x;
    else xu = x;
    if (xu <= xl) {
      xu = 0.5 * (xu + xl);
      xl = xu;
      break;
    }
    if (!ex_finite(f) || df < 1e-13) break;
    xp = x - f / df;
    if (fabs(xp - x) <= 1e-15 * MAXV(1.0, fabs(xp))) break;
    if (xp >= xu) x = MINV(0.05 * x + 0.95 * xu, xu);
    else if (xp <= xl) x = MAXV(0.05 * x + 0.95 * xl, xl);
    else x = xp;
  }
  if (i < 20) return ex_clip(x, xl, xu);
  return ex_bisect(v0, xl, xu, x);
}
static double ex_heur_primal(const double *v0, double *vp) { /* :160-182 */
  const double t0 = v0[2], s0 = v0[1], r0 = v0[0];
  double d;
  vp[2] = MAXV(t0, 0.0); vp[1] = 0; vp[0] = MINV(r0, 0.0);
  d = ex_dist_sq(v0, vp);
  if (s0 > 0.0) {
    const double tp = MAXV(t0, s0 * exp(r0 / s0)), nd = (tp - t0) * (tp - t0);
    if (nd < d) { vp[2] = tp; vp[1] = s0; vp[0] = r0; d = nd; }
  }
  return d;
}
static double ex_heur_polar(const double *v0, double *vd) { /* :185-207 */
  const double t0 = v0[2], s0 = v0[1], r0 = v0[0];
  double d;
  vd[2] = MINV(t0, 0.0); vd[1] = MINV(s0, 0.0); vd[0] = 0;
  d = ex_dist_sq(v0, vd);
  if (r0 > 0.0) {
    const double td = MINV(t0, -r0 * exp(s0 / r0 - 1.0)), nd = (t0 - td) * (t0 - td);
    if (nd < d) { vd[2] = td; vd[1] = s0; vd[0] = r0; d = nd; }
  }
  return d;
}
static double ex_ppsi(const double *v0) { /* :209-220 */
  const double s0 = v0[1], r0 = v0[0], q = sqrt(r0 * r0 + s0 * s0 - r0 * s0);
  const double psi = r0 > s0 ? (r0 - s0 + q) / r0 : -s0 / (r0 - s0 - q);
  return ((psi - 1.0) * r0 + s0) / (psi * (psi - 1.0) + 1.0);
}
static double ex_pomega(double rho) { /* :222-229 */
  double v = exp(rho) / (rho * (rho - 1.0) + 1.0);
  if (rho < 2.0) v = MINV(v, exp(2.0) / 3.0);
  return v;
}
static double ex_dpsi(const double *v0) { /* :231-242 */
  const double s0 = v0[1], r0 = v0[0], q = sqrt(r0 * r0 + s0 * s0 - r0 * s0);
  const double psi = s0 > r0 ? (r0 - q) / s0 : (r0 - s0) / (r0 + q);
  return (r0 - psi * s0) / (psi * (psi - 1.0) + 1.0);
}
static double ex_domega(double rho) { /* :244-251 */
  double v = -exp(-rho) / (rho * (rho - 1.0) + 1.0);
  if (rho > -1.0) v = MAXV(v, -exp(1.0) / 3.0);
  return v;
}
static void ex_bracket(const double *v0, double pd, double dd, double *lo, double *up) { /* :254-317 */
  const double t0 = v0[2], s0 = v0[1], r0 = v0[0];
  double baselow = -EXP_INF, baseupr = EXP_INF, low = -EXP_INF, upr = EXP_INF;
  const double ms = MINV(s0, 0.0), mr = MINV(r0, 0.0);
  const double Dp = sqrt(MAXV(pd - ms * ms, 0.0)), Dd = sqrt(MAXV(dd - mr * mr, 0.0));
  if (t0 > 0.0) low = MAXV(low, log(t0 / ex_ppsi(v0)));
  else if (t0 < 0.0) upr = MINV(upr, -log(-t0 / ex_dpsi(v0)));
  if (r0 > 0.0) {
    double tpu, val, sgn;
    baselow = 1.0 - s0 / r0;
    low = MAXV(low, baselow);
    tpu = MAXV(1e-12, MINV(Dd, Dp + t0));
    val = r0 * ex_pomega(low);
    sgn = val < 0 ? -1.0 : 1.0;
    upr = MINV(upr, MAXV(low, baselow + ex_safediv(tpu, fabs(val)) * sgn));
  }
  if (s0 > 0.0) {
    double tdl, val, sgn;
    baseupr = r0 / s0;
    upr = MINV(upr, baseupr);
    tdl = -MAXV(1e-12, MINV(Dp, Dd - t0));
    val = s0 * ex_domega(upr);
    sgn = val < 0 ? -1.0 : 1.0;
    low = MAXV(low, MINV(upr, baseupr - ex_safediv(tdl, fabs(val)) * sgn));
  }
  low = ex_clip(MINV(low, upr), baselow, baseupr);
  upr = ex_clip(MAXV(low, upr), baselow, baseupr);
  if (low != upr) {
    double fl, fu;
    ex_h(v0, low, &fl, NULL);
    ex_h(v0, upr, &fu, NULL);
    if (fl * fu > 0.0) {
      if (fabs(fl) < fabs(fu)) upr = low;
      else low = upr;
    }
  }
  *lo = low;
  *up = upr;
}
static double ex_sol_primal(const double *v0, double rho, double *vp) { /* :320-342 */
  const double lin = (rho - 1.0) * v0[0] + v0[1], er = exp(rho);
  if (lin > 0.0 && ex_finite(er)) {
    const double q = rho * (rho - 1.0) + 1.0;
    vp[2] = er * lin / q; vp[1] = lin / q; vp[0] = rho * lin / q;
    return ex_dist_sq(vp, v0);
  }
  vp[2] = EXP_INF; vp[1] = 0; vp[0] = 0;
  return EXP_INF;
}
static double ex_sol_polar(const double *v0, double rho, double *vd) { /* :345-367 */
  const double lin = v0[0] - rho * v0[1], er = exp(-rho);
  if (lin > 0.0 && ex_finite(er)) {
    const double q = rho * (rho - 1.0) + 1.0;
    vd[2] = -er * lin / q; vd[1] = (1.0 - rho) * lin / q; vd[0] = lin / q;
    return ex_dist_sq(v0, vd);
  }
  vd[2] = -EXP_INF; vd[1] = 0; vd[0] = 0;
  return EXP_INF;
}
static void cone_exp(double *v0, int primal) { /* exp_cone.c:373-441 */
  const double TOL = 1e-8;
  double vp[3], vd[3], vh[3], xl, xh, pd, dd, err;
  int opt, i;
  if (!primal) for (i = 0; i < 3; ++i) v0[i] = -v0[i];
  pd = ex_heur_primal(v0, vp);
  dd = ex_heur_polar(v0, vd);
  err = fabs(vp[0] + vd[0] - v0[0]);
  err = MAXV(err, fabs(vp[1] + vd[1] - v0[1]));
  err = MAXV(err, fabs(vp[2] + vd[2] - v0[2]));
  opt = v0[1] <= 0.0 && v0[0] <= 0.0;
  opt = opt || MINV(pd, dd) <= TOL * TOL;
  opt = opt || (err <= TOL && (vp[0] * vd[0] + vp[1] * vd[1] + vp[2] * vd[2]) <= TOL);
  if (!opt) {
    double rho;
    ex_bracket(v0, pd, dd, &xl, &xh);
    rho = ex_newton(v0, xl, xh, 0.5 * (xl + xh));
    if (primal) {
      if (ex_sol_primal(v0, rho, vh) <= pd) memcpy(vp, vh, sizeof vp);
    } else {
      if (ex_sol_polar(v0, rho, vh) <= dd) memcpy(vd, vh, sizeof vd);
    }
  }
  if (primal) memcpy(v0, vp, sizeof vp);
  else for (i = 0; i < 3; ++i) v0[i] = -vd[i]; /* polar -> dual */
}

/* ---- power cone: src/cones.c:1284-1335 (Newton on r, <= 20 steps) ---- */
static double pw_x(double r, double xh, double rh, double a) {
  const double x = 0.5 * (xh + sqrt(xh * xh + 4 * a * (rh - r) * r));
  return MAXV(x, 1e-12);
}
static void cone_pow(double *v, double a) {
  const double PTOL = 1e-9, xh = v[0], yh = v[1], rh = fabs(v[2]);
  double x = 0, y = 0, r;
  int i;
  if (xh >= 0 && yh >= 0 && PTOL + pow(xh, a) * pow(yh, 1 - a) >= rh) return;
  if (xh <= 0 && yh <= 0 && PTOL + pow(-xh, a) * pow(-yh, 1 - a) >= rh * pow(a, a) * pow(1 - a, 1 - a)) {
    v[0] = v[1] = v[2] = 0;
    return;
  }
  r = rh / 2;
  for (i = 0; i < 20; ++i) {
    double xa, y1a, f, dxdr, dydr, fp;
    x = pw_x(r, xh, rh, a);
    y = pw_x(r, yh, rh, 1 - a);
    xa = pow(x, a);
    y1a = pow(y, 1 - a);
    f = xa * y1a - r;
    if (fabs(f) < PTOL) break;
    dxdr = a * (rh - 2 * r) / (2 * x - xh);
    dydr = (1 - a) * (rh - 2 * r) / (2 * y - yh);
    fp = xa * y1a * (a * dxdr / x + (1 - a) * dydr / y) - 1;
    r = MAXV(r - f / fp, 0.0);
    r = MINV(r, rh);
  }
  v[0] = x;
  v[1] = y;
  v[2] = v[2] < 0 ? -r : r;
}

struct OrCone {
  int m, z, l, bsize, qsize, ssize;
  int cssize, ep, ed, psize; /* complex PSD, exponential (primal, dual), power cones */
  int *cs;
  double *pw;
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
    if (!Db) { /* cones.c:1561: normalize_box_cone only runs when a scaling exists */
      c->bu[j] = bu[j];
      c->bl[j] = bl[j];
      continue;
    }
    c->bu[j] = bu[j] >= 1e15 ? INFINITY : bu[j] * f;
    c->bl[j] = bl[j] <= -1e15 ? -INFINITY : bl[j] * f;
  }
  c->box_t_warm_start = 1.;
  c->scratch = (double *)calloc(m > 0 ? m : 1, sizeof(double));
  return c;
}
/* the cones that follow the PSD blocks in the row order of cones.c:1340-1394: complex PSD,
 * exponential (primal then dual, 3 rows each), power (3 rows each; negative parameter = dual) */
void or_cone_set_extra(OrCone *c, int cssize, const int *cs, int ep, int ed, int psize, const double *pw) {
  c->cssize = cssize; c->ep = ep; c->ed = ed; c->psize = psize;
  free(c->cs); free(c->pw);
  c->cs = (int *)calloc(cssize > 0 ? cssize : 1, sizeof(int));
  c->pw = (double *)calloc(psize > 0 ? psize : 1, sizeof(double));
  if (cssize) memcpy(c->cs, cs, cssize * sizeof(int));
  if (psize) memcpy(c->pw, pw, psize * sizeof(double));
}
void or_cone_free(OrCone *c) {
  if (!c) return;
  free(c->cs); free(c->pw);
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
  for (i = 0; i < c->cssize; ++i) {
    cone_cpsd(x + count, c->cs[i]);
    count += c->cs[i] * c->cs[i];
  }
  for (i = 0; i < c->ep + c->ed; ++i) {
    cone_exp(x + count, i < c->ep);
    count += 3;
  }
  for (i = 0; i < c->psize; ++i) {
    double *v = x + count;
    if (c->pw[i] >= 0) {
      cone_pow(v, c->pw[i]);
    } else { /* dual power cone through Moreau, cones.c:1427-1441 */
      double w[3];
      w[0] = -v[0]; w[1] = -v[1]; w[2] = -v[2];
      cone_pow(w, -c->pw[i]);
      v[0] += w[0]; v[1] += w[1]; v[2] += w[2];
    }
    count += 3;
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
  int count = c->z + c->l + c->bsize, i, j;
  const int n3 = c->ep + c->ed + c->psize, nc = c->qsize + c->ssize + c->cssize + n3;
  for (i = 0; i < nc; ++i) { /* set_cone_boundaries, cones.c:386-424: q, s(s+1)/2, cs^2, then 3-row cones */
    int delta;
    if (i < c->qsize) delta = c->q[i];
    else if (i < c->qsize + c->ssize) delta = c->s[i - c->qsize] * (c->s[i - c->qsize] + 1) / 2;
    else if (i < c->qsize + c->ssize + c->cssize) delta = c->cs[i - c->qsize - c->ssize] * c->cs[i - c->qsize - c->ssize];
    else delta = 3;
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
  return or_solve_ext(m, n, Ap, Ai, Ax_in, b_in, c_in, z, l, bsize, bl, bu, qsize, q, ssize, s, 0, NULL, 0, 0, 0, NULL,
                      st, xo, yo, so, info);
}

/* the same loop over every cone type (complex PSD, exponential, power after the PSD blocks) */
int or_solve_ext(int m, int n, const int *Ap, const int *Ai, const double *Ax_in, const double *b_in,
                 const double *c_in, int z, int l, int bsize, const double *bl, const double *bu, int qsize,
                 const int *q, int ssize, const int *s, int cssize, const int *cs, int ep, int ed, int psize,
                 const double *pw, const OrSettings *st, double *xo, double *yo, double *so, OrInfo *info) {
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
  or_cone_set_extra(cone, cssize, cs, ep, ed, psize, pw);
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
    or_cone_set_extra(cone, cssize, cs, ep, ed, psize, pw);
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
