// aa_host.cpp -- host-side Anderson acceleration of the ADMM fixed-point map.
//
// Stays on the host by design (BASELINE.json north_star); the driver ships v and
// v_prev over PCIe only on the iterations that call in here.  Restates the
// algorithm of reference src/aa.c without LAPACK:
//   aa_init :657-820, aa_apply :822-854, aa_safeguard :856-899, aa_reset :934-967
//   update_accel_params :340-391  (S, D, Y columns; g = x - f; cached column norms)
//   compute_regularization :253-270  r = reg * ||A||_F ||Y||_F  (reg<0: pinned |reg|)
//   solve :422-655  pivoted QR of [A; sqrt(r) I] (A = S type-I, Y type-II), rank
//                   truncation at len*eps*|R11|, Q'[g;0], reduced solve (type-I: LU of
//                   the top block of Q'[Y_piv; sqrt(r) e_piv]; type-II: R u = c) with
//                   iterative refinement, weight-norm cap, f -= D gamma, relaxation.
// The dense kernels the reference takes from LAPACK (geqp3, ormqr, gesv, getrs,
// trsv, trmv) are small here (<= mem columns) and written out below: Householder
// QR with column pivoting and norm downdating, reflector application, LU with
// partial pivoting.
#include "scs_host.h"
#include "aa_small.h"
#include <algorithm>
#include <cfloat>
#include <cmath>

namespace scsamd {

struct AaHost {
  int type1 = 1, mem = 0, min_len = 0, dim = 0, iter = 0, success = 0, ir_max_steps = 0;
  real relaxation = 1, regularization = 0, safeguard_factor = 1, max_weight_norm = 0;
  real norm_g = 0;
  std::vector<real> x, f, g, g_prev, Y, S, D, nrm_s_col, nrm_y_col;
  std::vector<real> A_aug, B_aug, c_aug, tau, W, W_orig, gamma_red, c_top, ir_res, work, x_work, colnrm, colnrm0;
  std::vector<int> jpvt, ipiv;
  AaStats st;
};

AaHost *aa_host_init(int dim, int mem, int min_len, int type1, real regularization, real relaxation,
                     real safeguard_factor, real max_weight_norm, int ir_max_steps) {
  const int memc = std::min(mem, dim);
  if (dim <= 0 || mem < 0 || !std::isfinite((double)regularization) || relaxation < 0 || relaxation > 2 ||
      safeguard_factor < 0 || max_weight_norm <= 0 || ir_max_steps < 0 || (memc > 0 && min_len < 1)) {
    printf("Invalid AA parameters.\n");
    return nullptr;
  }
  AaHost *a = new AaHost();
  a->type1 = type1;
  a->dim = dim;
  a->mem = memc;
  a->min_len = memc > 0 ? std::min(min_len, memc) : 0;
  a->regularization = regularization;
  a->relaxation = relaxation;
  a->safeguard_factor = safeguard_factor;
  a->max_weight_norm = max_weight_norm;
  a->ir_max_steps = ir_max_steps;
  memset(&a->st, 0, sizeof a->st);
  a->st.last_aa_norm = (real)NAN;
  if (memc <= 0) return a;
  try {
    const size_t d = (size_t)dim, m = (size_t)memc, aug = d + m;
    a->x.assign(d, 0); a->f.assign(d, 0); a->g.assign(d, 0); a->g_prev.assign(d, 0);
    a->Y.assign(d * m, 0); a->S.assign(d * m, 0); a->D.assign(d * m, 0);
    a->nrm_s_col.assign(m, 0); a->nrm_y_col.assign(m, 0);
    a->A_aug.assign(aug * m, 0); a->c_aug.assign(aug, 0); a->tau.assign(m, 0);
    a->jpvt.assign(m, 0); a->colnrm.assign(m, 0); a->colnrm0.assign(m, 0);
    a->gamma_red.assign(m, 0); a->c_top.assign(m, 0); a->ir_res.assign(m, 0);
    if (type1) {
      a->B_aug.assign(aug * m, 0); a->W.assign(m * m, 0); a->W_orig.assign(m * m, 0); a->ipiv.assign(m, 0);
    }
    a->work.assign(std::max(d, m), 0);
    if (relaxation != (real)1.0) a->x_work.assign(d, 0);
  } catch (const std::bad_alloc &) {
    printf("Failed to allocate memory for AA.\n");
    delete a;
    return nullptr;
  }
  return a;
}

void aa_host_reset(AaHost *a) { // aa.c:934-967
  if (!a) return;
  a->iter = 0;
  a->success = 0;
  a->norm_g = 0;
  std::fill(a->nrm_s_col.begin(), a->nrm_s_col.end(), (real)0);
  std::fill(a->nrm_y_col.begin(), a->nrm_y_col.end(), (real)0);
}

void aa_host_finish(AaHost *a) { delete a; }

void aa_host_stats(const AaHost *a, AaStats *out) {
  *out = a->st;
  out->iter = a->iter;
}

// ---- dense kernels on tall-skinny column-major matrices ------------------------------
// Householder QR with column pivoting of the rows x len matrix A (leading dim = rows).
// On exit R is in the upper triangle, reflector k is [1; A[k+1:, k]] with scalar tau[k],
// jpvt[k] = original index of the column now in position k.
static void qr_pivoted(real *A, long rows, int len, int *jpvt, real *tau, real *cn, real *cn0) {
  for (int j = 0; j < len; ++j) {
    jpvt[j] = j;
    cn[j] = cn0[j] = nrm2(A + (size_t)j * rows, rows);
  }
  const real tol3z = std::sqrt((real)(sizeof(real) == 8 ? DBL_EPSILON : FLT_EPSILON));
  for (int k = 0; k < len; ++k) {
    int piv = k;
    for (int j = k + 1; j < len; ++j)
      if (cn[j] > cn[piv]) piv = j;
    if (piv != k) {
      real *ck = A + (size_t)k * rows, *cp = A + (size_t)piv * rows;
      for (long i = 0; i < rows; ++i) std::swap(ck[i], cp[i]);
      std::swap(jpvt[k], jpvt[piv]);
      cn[piv] = cn[k];
      cn0[piv] = cn0[k];
    }
    real *v = A + (size_t)k * rows;
    const real alpha = v[k];
    const real xnorm = nrm2(v + k + 1, rows - k - 1);
    if (xnorm == 0) {
      tau[k] = 0;
    } else {
      const real beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
      tau[k] = (beta - alpha) / beta;
      const real sc = (real)1 / (alpha - beta);
      for (long i = k + 1; i < rows; ++i) v[i] *= sc;
      v[k] = beta;
    }
    for (int j = k + 1; j < len; ++j) { // apply H_k to the trailing columns, downdate norms
      real *c = A + (size_t)j * rows;
      if (tau[k] != 0) {
        real w = c[k];
        for (long i = k + 1; i < rows; ++i) w += v[i] * c[i];
        w *= tau[k];
        c[k] -= w;
        for (long i = k + 1; i < rows; ++i) c[i] -= w * v[i];
      }
      if (cn[j] != 0) {
        real t = std::fabs(c[k]) / cn[j];
        t = std::max((real)0, (1 + t) * (1 - t));
        const real t2 = t * (cn[j] / cn0[j]) * (cn[j] / cn0[j]);
        if (t2 <= tol3z) {
          cn[j] = nrm2(c + k + 1, rows - k - 1);
          cn0[j] = cn[j];
        } else {
          cn[j] *= std::sqrt(t);
        }
      }
    }
  }
}

// c <- H_{nref-1} ... H_0 c  (= Q' c restricted to the first nref reflectors)
static void apply_qt(const real *A, long rows, int nref, const real *tau, real *c) {
  for (int k = 0; k < nref; ++k) {
    if (tau[k] == 0) continue;
    const real *v = A + (size_t)k * rows;
    real w = c[k];
    for (long i = k + 1; i < rows; ++i) w += v[i] * c[i];
    w *= tau[k];
    c[k] -= w;
    for (long i = k + 1; i < rows; ++i) c[i] -= w * v[i];
  }
}

// ---- the acceleration step -----------------------------------------------------------
static real aa_solve(real *f, AaHost *a, int len) {
  const long dim = a->dim, aug = dim + a->mem;
  const int mem = a->mem;
  const real *A_src = a->type1 ? a->S.data() : a->Y.data();
  real *gamma = a->work.data();
  real r;
  if (a->regularization > 0) {
    const real ny = frob_from_cols(a->nrm_y_col);
    const real na = a->type1 ? frob_from_cols(a->nrm_s_col) : ny;
    r = a->regularization * na * ny;
  } else if (a->regularization < 0) {
    r = -a->regularization;
  } else {
    r = 0;
  }
  const real sqrt_r = r > 0 ? std::sqrt(r) : (real)0;
  // [A; sqrt(r) I] column by column (aa.c:272-291)
  for (int i = 0; i < len; ++i) {
    real *col = a->A_aug.data() + (size_t)i * aug;
    memcpy(col, A_src + (size_t)i * dim, dim * sizeof(real));
    memset(col + dim, 0, mem * sizeof(real));
    col[dim + i] = sqrt_r;
  }
  qr_pivoted(a->A_aug.data(), aug, len, a->jpvt.data(), a->tau.data(), a->colnrm.data(), a->colnrm0.data());
  int rank = 0, info = 0;
  {
    const real r11 = std::fabs(a->A_aug[0]);
    if (r11 > 0) {
      const real tol = r11 * (real)len * (real)(sizeof(real) == 8 ? DBL_EPSILON : FLT_EPSILON);
      for (rank = 0; rank < len; ++rank)
        if (std::fabs(a->A_aug[(size_t)rank * aug + rank]) < tol) break;
    }
    if (rank == 0) info = 1;
  }
  if (info == 0) {
    memcpy(a->c_aug.data(), a->g.data(), dim * sizeof(real));
    memset(a->c_aug.data() + dim, 0, mem * sizeof(real));
    apply_qt(a->A_aug.data(), aug, rank, a->tau.data(), a->c_aug.data());
    memcpy(a->c_top.data(), a->c_aug.data(), rank * sizeof(real));
    if (a->type1) {
      for (int i = 0; i < rank; ++i) {
        const int piv = a->jpvt[i];
        real *col = a->B_aug.data() + (size_t)i * aug;
        memcpy(col, a->Y.data() + (size_t)piv * dim, dim * sizeof(real));
        memset(col + dim, 0, mem * sizeof(real));
        col[dim + piv] = sqrt_r;
        apply_qt(a->A_aug.data(), aug, rank, a->tau.data(), col);
      }
      for (int i = 0; i < rank; ++i) {
        memcpy(&a->W[(size_t)i * mem], a->B_aug.data() + (size_t)i * aug, rank * sizeof(real));
        memcpy(&a->W_orig[(size_t)i * mem], &a->W[(size_t)i * mem], rank * sizeof(real));
      }
      memcpy(a->gamma_red.data(), a->c_top.data(), rank * sizeof(real));
      info = lu_factor(a->W.data(), rank, mem, a->ipiv.data());
      if (info == 0) {
        lu_solve(a->W.data(), rank, mem, a->ipiv.data(), a->gamma_red.data());
        real prev = 0;
        for (int k = 0; k < a->ir_max_steps; ++k) { // iterative refinement, aa.c:530-552
          for (int i = 0; i < rank; ++i) {
            real s = a->c_top[i];
            for (int j = 0; j < rank; ++j) s -= a->W_orig[i + (size_t)j * mem] * a->gamma_red[j];
            a->ir_res[i] = s;
          }
          lu_solve(a->W.data(), rank, mem, a->ipiv.data(), a->ir_res.data());
          const real dn = nrm2(a->ir_res.data(), rank);
          for (int i = 0; i < rank; ++i) a->gamma_red[i] += a->ir_res[i];
          if (k > 0 && dn >= (real)0.5 * prev) break;
          prev = dn;
        }
      }
    } else {
      memcpy(a->gamma_red.data(), a->c_top.data(), rank * sizeof(real));
      upper_solve(a->A_aug.data(), aug, rank, a->gamma_red.data());
      real prev = 0;
      for (int k = 0; k < a->ir_max_steps; ++k) { // aa.c:566-585
        for (int i = 0; i < rank; ++i) {
          real s = 0;
          for (int j = i; j < rank; ++j) s += a->A_aug[i + (size_t)j * aug] * a->gamma_red[j];
          a->ir_res[i] = a->c_top[i] - s;
        }
        upper_solve(a->A_aug.data(), aug, rank, a->ir_res.data());
        const real dn = nrm2(a->ir_res.data(), rank);
        for (int i = 0; i < rank; ++i) a->gamma_red[i] += a->ir_res[i];
        if (k > 0 && dn >= (real)0.5 * prev) break;
        prev = dn;
      }
    }
    if (info == 0) {
      for (int i = 0; i < len; ++i) gamma[i] = 0;
      for (int i = 0; i < rank; ++i) gamma[a->jpvt[i]] = a->gamma_red[i];
    }
  }
  real aa_norm = info == 0 ? nrm2(gamma, len) : (real)-1.0;
  a->st.last_rank = rank;
  a->st.last_regularization = r;
  a->st.last_aa_norm = (info == 0 && std::isfinite((double)aa_norm)) ? aa_norm : (real)NAN;
  if (info != 0 || !std::isfinite((double)aa_norm) || aa_norm >= a->max_weight_norm) {
    if (rank == 0) a->st.n_reject_rank0++;
    else if (info != 0) a->st.n_reject_lapack++;
    else if (!std::isfinite((double)aa_norm)) a->st.n_reject_nonfinite++;
    else a->st.n_reject_weight_cap++;
    a->success = 0;
    aa_host_reset(a);
    if (!std::isfinite((double)aa_norm)) aa_norm = -1.0;
    return aa_norm < 0 ? aa_norm : -aa_norm;
  }
  // f -= D gamma
  for (int j = 0; j < len; ++j) {
    const real gj = gamma[j];
    if (gj == 0) continue;
    const real *dc = a->D.data() + (size_t)j * dim;
    for (long i = 0; i < dim; ++i) f[i] -= dc[i] * gj;
  }
  if (a->relaxation != (real)1.0) { // aa.c:393-410
    for (int j = 0; j < len; ++j) {
      const real gj = gamma[j];
      const real *sc = a->S.data() + (size_t)j * dim;
      for (long i = 0; i < dim; ++i) a->x_work[i] -= sc[i] * gj;
    }
    const real om = (real)1. - a->relaxation;
    for (long i = 0; i < dim; ++i) f[i] = a->relaxation * f[i] + om * a->x_work[i];
  }
  a->success = 1;
  return aa_norm;
}

real aa_host_apply(real *f, const real *x, AaHost *a) { // aa.c:822-854
  real aa_norm = 0;
  const int len = std::min(a->iter, a->mem);
  const long dim = a->dim;
  a->success = 0;
  if (a->mem <= 0) return aa_norm;
  if (a->iter == 0) { // seed (init_accel_params, aa.c:293-307)
    memcpy(a->x.data(), x, dim * sizeof(real));
    memcpy(a->f.data(), f, dim * sizeof(real));
    for (long i = 0; i < dim; ++i) a->g_prev[i] = x[i] - f[i];
    a->iter++;
    return aa_norm;
  }
  { // update_accel_params, aa.c:340-391
    const int idx = (a->iter - 1) % a->mem;
    real *sc = a->S.data() + (size_t)idx * dim, *dc = a->D.data() + (size_t)idx * dim,
         *yc = a->Y.data() + (size_t)idx * dim;
    for (long i = 0; i < dim; ++i) {
      sc[i] = x[i] - a->x[i];
      dc[i] = f[i] - a->f[i];
      const real gi = x[i] - f[i];
      a->g[i] = gi;
      yc[i] = gi - a->g_prev[i];
    }
    a->nrm_s_col[idx] = nrm2(sc, dim);
    a->nrm_y_col[idx] = nrm2(yc, dim);
    memcpy(a->x.data(), x, dim * sizeof(real));
    memcpy(a->f.data(), f, dim * sizeof(real));
    memcpy(a->g_prev.data(), a->g.data(), dim * sizeof(real));
    if (!a->x_work.empty()) memcpy(a->x_work.data(), x, dim * sizeof(real));
    a->norm_g = nrm2(a->g.data(), dim);
  }
  if (a->iter >= a->min_len) {
    aa_norm = aa_solve(f, a, len);
    if (aa_norm > 0) a->st.n_accept++;
  }
  a->iter++;
  return aa_norm;
}

int aa_host_safeguard(real *f_new, real *x_new, AaHost *a) { // aa.c:856-899
  if (a->mem <= 0 || !a->success) return 0;
  a->success = 0;
  const long dim = a->dim;
  for (long i = 0; i < dim; ++i) a->work[i] = x_new[i] - f_new[i];
  const real nd = nrm2(a->work.data(), dim);
  if (nd > a->safeguard_factor * a->norm_g) {
    memcpy(f_new, a->f.data(), dim * sizeof(real));
    memcpy(x_new, a->x.data(), dim * sizeof(real));
    a->st.n_safeguard_reject++;
    aa_host_reset(a);
    return -1;
  }
  return 0;
}

} // namespace scsamd

// ---- C ABI exposure of the host AA (so the CPU test-suite can pin it against the
// reference's aa_init/aa_apply/aa_safeguard without a GPU) --------------------------------
using namespace scsamd;
extern "C" {
void *scs_amd_aa_init(scs_int dim, scs_int mem, scs_int min_len, scs_int type1, scs_float regularization,
                      scs_float relaxation, scs_float safeguard_factor, scs_float max_weight_norm,
                      scs_int ir_max_steps) {
  return aa_host_init(dim, mem, min_len, type1, regularization, relaxation, safeguard_factor, max_weight_norm,
                      ir_max_steps);
}
scs_float scs_amd_aa_apply(scs_float *f, const scs_float *x, void *a) { return aa_host_apply(f, x, (AaHost *)a); }
scs_int scs_amd_aa_safeguard(scs_float *f_new, scs_float *x_new, void *a) {
  return aa_host_safeguard(f_new, x_new, (AaHost *)a);
}
void scs_amd_aa_reset(void *a) { aa_host_reset((AaHost *)a); }
void scs_amd_aa_finish(void *a) { aa_host_finish((AaHost *)a); }
void scs_amd_aa_get_stats(const void *a, AaStats *out) { aa_host_stats((const AaHost *)a, out); }
}
