// equilibrate.cpp -- host-side data equilibration (setup time, runs once).
//
// Restates what the reference does before the linear-system backend ever sees
// the data, so that B2 hands the device the same normalized problem:
//   normalize_a_p      linsys/scs_matrix.c:433-496   25 Ruiz passes + 1 L2 pass
//   compute_ruiz_mats  :236-307    D = 1/sqrt(row inf-norm), E = 1/sqrt(col inf-norm)
//   compute_l2_mats    :309-368    same with 2-norms; rows averaged per cone
//   rescale            :370-407    A <- D A E, P <- E P E, accumulate D, E
//   enforce_cone_boundaries  src/cones.c:366-379   D constant inside each cone
//   normalize_b_c      src/normalize.c:33-61
//   (un)normalize_sol  src/normalize.c:64-91
// The matrices are host CSC copies owned by the workspace; everything is O(nnz)
// streaming work, a candidate for the device later.
#include "scs_host.h"
#include <algorithm>
#include <cmath>

namespace scsamd {

static const double MIN_NORM_FACTOR = 1e-4, MAX_NORM_FACTOR = 1e4; // scs_matrix.c:13-14
static const int NUM_RUIZ_PASSES = 25, NUM_L2_PASSES = 1;           // scs_matrix.c:15-16

static inline real limit_scale(real x) { // scs_matrix.c:229-234
  x = x < (real)MIN_NORM_FACTOR ? (real)1.0 : x;
  x = x > (real)MAX_NORM_FACTOR ? (real)MAX_NORM_FACTOR : x;
  return x;
}
static inline real safe_div_pos(real x, real y) { // glbopts.h:194-196
  return y < (real)1e-18 ? x / (real)1e-18 : x / y;
}

// Cone boundaries as the reference builds them (src/cones.c:386-424): the first
// segment (zero + pos + box rows) is left row-wise, every further cone is one
// segment whose rows must share a single scale factor.
std::vector<int> cone_segments(const ScsCone *k) {
  std::vector<int> b;
  b.push_back(k->z + k->l + k->bsize);
  for (int i = 0; i < k->qsize; ++i) b.push_back(k->q[i]);
  for (int i = 0; i < k->ssize; ++i) b.push_back(k->s[i] * (k->s[i] + 1) / 2);
  for (int i = 0; i < k->cssize; ++i) b.push_back(k->cs[i] * k->cs[i]);
  for (int i = 0; i < k->ep + k->ed + k->psize; ++i) b.push_back(3);
  return b;
}

enum Agg { AGG_MAX, AGG_MEAN };
static void aggregate_over_cones(const std::vector<int> &seg, std::vector<real> &v, Agg how) {
  size_t pos = (size_t)seg[0];
  for (size_t c = 1; c < seg.size(); ++c) {
    const int len = seg[c];
    real w = 0;
    if (how == AGG_MAX) {
      for (int j = 0; j < len; ++j) w = std::max(w, (real)std::fabs(v[pos + j]));
    } else if (len > 0) {
      for (int j = 0; j < len; ++j) w += v[pos + j];
      w /= (real)len;
    }
    for (int j = 0; j < len; ++j) v[pos + j] = w;
    pos += (size_t)len;
  }
}

static void ruiz_pass(const HostCsc *P, const HostCsc &A, const std::vector<int> &seg,
                      std::vector<real> &Dt, std::vector<real> &Et) {
  const int m = A.m, n = A.n;
  std::fill(Dt.begin(), Dt.end(), (real)0);
  for (int j = 0; j < n; ++j)
    for (eoff k = A.p[j]; k < A.p[j + 1]; ++k) Dt[A.i[k]] = std::max(Dt[A.i[k]], (real)std::fabs(A.x[k]));
  aggregate_over_cones(seg, Dt, AGG_MAX);
  for (int i = 0; i < m; ++i) Dt[i] = safe_div_pos((real)1, std::sqrt(limit_scale(Dt[i])));
  std::fill(Et.begin(), Et.end(), (real)0);
  if (P)
    for (int j = 0; j < n; ++j)
      for (eoff k = P->p[j]; k < P->p[j + 1]; ++k) {
        const int i = P->i[k];
        const real w = std::fabs(P->x[k]);
        Et[j] = std::max(Et[j], w);
        if (i != j) Et[i] = std::max(Et[i], w);
      }
  for (int j = 0; j < n; ++j) {
    real cn = 0;
    for (eoff k = A.p[j]; k < A.p[j + 1]; ++k) cn = std::max(cn, (real)std::fabs(A.x[k]));
    Et[j] = std::max(Et[j], cn);
    Et[j] = safe_div_pos((real)1, std::sqrt(limit_scale(Et[j])));
  }
}

static void l2_pass(const HostCsc *P, const HostCsc &A, const std::vector<int> &seg,
                    std::vector<real> &Dt, std::vector<real> &Et) {
  const int m = A.m, n = A.n;
  std::fill(Dt.begin(), Dt.end(), (real)0);
  for (int j = 0; j < n; ++j)
    for (eoff k = A.p[j]; k < A.p[j + 1]; ++k) Dt[A.i[k]] += A.x[k] * A.x[k];
  for (int i = 0; i < m; ++i) Dt[i] = std::sqrt(Dt[i]);
  aggregate_over_cones(seg, Dt, AGG_MEAN);
  for (int i = 0; i < m; ++i) Dt[i] = safe_div_pos((real)1, std::sqrt(limit_scale(Dt[i])));
  std::fill(Et.begin(), Et.end(), (real)0);
  if (P)
    for (int j = 0; j < n; ++j)
      for (eoff k = P->p[j]; k < P->p[j + 1]; ++k) {
        const int i = P->i[k];
        const real w = P->x[k] * P->x[k];
        Et[j] += w;
        if (i != j) Et[i] += w;
      }
  for (int j = 0; j < n; ++j) {
    real ss = 0;
    for (eoff k = A.p[j]; k < A.p[j + 1]; ++k) ss += A.x[k] * A.x[k];
    Et[j] += ss;
    Et[j] = safe_div_pos((real)1, std::sqrt(limit_scale(std::sqrt(Et[j]))));
  }
}

static void apply_scaling(HostCsc *P, HostCsc &A, const std::vector<real> &Dt, const std::vector<real> &Et,
                          Scaling &sc) {
  for (int j = 0; j < A.n; ++j) {
    const real ej = Et[j];
    for (eoff k = A.p[j]; k < A.p[j + 1]; ++k) A.x[k] *= Dt[A.i[k]] * ej;
  }
  if (P)
    for (int j = 0; j < P->n; ++j) {
      const real ej = Et[j];
      for (eoff k = P->p[j]; k < P->p[j + 1]; ++k) P->x[k] *= Et[P->i[k]] * ej;
    }
  for (int i = 0; i < A.m; ++i) sc.D[i] *= Dt[i];
  for (int j = 0; j < A.n; ++j) sc.E[j] *= Et[j];
}

void equilibrate(HostCsc *P, HostCsc &A, const ScsCone *k, Scaling &sc) {
  sc.D.assign((size_t)A.m, (real)1);
  sc.E.assign((size_t)A.n, (real)1);
  sc.primal_scale = sc.dual_scale = 1;
  std::vector<real> Dt((size_t)A.m), Et((size_t)A.n);
  const std::vector<int> seg = cone_segments(k);
  for (int pass = 0; pass < NUM_RUIZ_PASSES; ++pass) {
    ruiz_pass(P, A, seg, Dt, Et);
    apply_scaling(P, A, Dt, Et, sc);
  }
  for (int pass = 0; pass < NUM_L2_PASSES; ++pass) {
    l2_pass(P, A, seg, Dt, Et);
    apply_scaling(P, A, Dt, Et, sc);
  }
}

// b <- sigma D b, c <- sigma E c with sigma = 1/clip(max(|Db|_inf, |Ec|_inf))
void normalize_b_c(Scaling &sc, real *b, real *c) {
  const size_t m = sc.D.size(), n = sc.E.size();
  real nb = 0, nc = 0;
  for (size_t j = 0; j < n; ++j) {
    c[j] *= sc.E[j];
    nc = std::max(nc, (real)std::fabs(c[j]));
  }
  for (size_t i = 0; i < m; ++i) {
    b[i] *= sc.D[i];
    nb = std::max(nb, (real)std::fabs(b[i]));
  }
  real sigma = std::max(nc, nb);
  sigma = sigma < (real)MIN_NORM_FACTOR ? (real)1 : sigma;
  sigma = sigma > (real)MAX_NORM_FACTOR ? (real)MAX_NORM_FACTOR : sigma;
  sigma = safe_div_pos((real)1, sigma);
  for (size_t j = 0; j < n; ++j) c[j] *= sigma;
  for (size_t i = 0; i < m; ++i) b[i] *= sigma;
  sc.primal_scale = sc.dual_scale = sigma;
}

void normalize_sol(const Scaling &sc, real *x, real *y, real *s) {
  for (size_t j = 0; j < sc.E.size(); ++j) x[j] /= (sc.E[j] / sc.dual_scale);
  for (size_t i = 0; i < sc.D.size(); ++i) {
    y[i] /= (sc.D[i] / sc.primal_scale);
    s[i] *= (sc.D[i] * sc.dual_scale);
  }
}

void un_normalize_sol(const Scaling &sc, real *x, real *y, real *s) {
  for (size_t j = 0; j < sc.E.size(); ++j) x[j] *= (sc.E[j] / sc.dual_scale);
  for (size_t i = 0; i < sc.D.size(); ++i) {
    y[i] *= (sc.D[i] / sc.primal_scale);
    s[i] /= (sc.D[i] * sc.dual_scale);
  }
}

// ---- input validation (reference linsys/scs_matrix.c:65-157) -----------------
int validate_csc(const ScsMatrix *M, int rows, int cols, bool upper_only, const char *name) {
  if (!M->x || !M->i || !M->p) {
    printf("data incompletely specified (%s)\n", name);
    return -1;
  }
  if (M->m != rows || M->n != cols) {
    printf("%s dimension mismatch\n", name);
    return -1;
  }
  if (M->p[0] != 0) {
    printf("%s->p[0] must be zero\n", name);
    return -1;
  }
  for (int j = 0; j < cols; ++j) {
    if (M->p[j + 1] < M->p[j]) {
      printf("%s->p not monotonically increasing\n", name);
      return -1;
    }
  }
  const long long nnz = M->p[cols];
  for (long long k = 0; k < nnz; ++k) {
    if (M->i[k] < 0 || M->i[k] >= rows) {
      printf("error: %s row index out of bounds\n", name);
      return -1;
    }
    if (!std::isfinite((double)M->x[k])) {
      printf("error: %s contains a non-finite value\n", name);
      return -1;
    }
  }
  if (upper_only)
    for (int j = 0; j < cols; ++j)
      for (eoff k = M->p[j]; k < M->p[j + 1]; ++k)
        if (M->i[k] > j) {
          printf("error: P is not upper triangular\n");
          return -1;
        }
  return 0;
}

} // namespace scsamd
