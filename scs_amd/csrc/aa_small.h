// aa_small.h -- the small dense pieces of Anderson acceleration that always run on
// the host (mem x mem systems, scalar norms); shared by the host AA (aa_host.cpp)
// and the device AA (aa_dev.hip).  Reference: src/aa.c:236-251 (Frobenius norm from
// cached column norms), :505-552 (gesv/getrs + refinement), :556-585 (trsv/trmv).
#pragma once
#include "scs_host.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace scsamd {

static inline real nrm2(const real *x, long n) { // scaled 2-norm (overflow safe, like BLAS nrm2)
  real scale = 0, ssq = 1;
  for (long i = 0; i < n; ++i) {
    if (x[i] != 0) {
      const real a = std::fabs(x[i]);
      if (scale < a) {
        ssq = 1 + ssq * (scale / a) * (scale / a);
        scale = a;
      } else {
        ssq += (a / scale) * (a / scale);
      }
    }
  }
  return scale * std::sqrt(ssq);
}

static inline real frob_from_cols(const std::vector<real> &c) { // aa.c:236-251
  real m = 0;
  for (real v : c) m = std::max(m, v);
  if (m == 0) return 0;
  real s = 0;
  for (real v : c) s += (v / m) * (v / m);
  return m * std::sqrt(s);
}

// LU with partial pivoting of the r x r matrix W (column major, leading dim ld); 0 on success
static inline int lu_factor(real *W, int r, int ld, int *ipiv) {
  for (int k = 0; k < r; ++k) {
    int p = k;
    for (int i = k + 1; i < r; ++i)
      if (std::fabs(W[i + (size_t)k * ld]) > std::fabs(W[p + (size_t)k * ld])) p = i;
    ipiv[k] = p;
    if (W[p + (size_t)k * ld] == 0) return k + 1;
    if (p != k)
      for (int j = 0; j < r; ++j) std::swap(W[k + (size_t)j * ld], W[p + (size_t)j * ld]);
    const real d = (real)1 / W[k + (size_t)k * ld];
    for (int i = k + 1; i < r; ++i) W[i + (size_t)k * ld] *= d;
    for (int j = k + 1; j < r; ++j) {
      const real wkj = W[k + (size_t)j * ld];
      for (int i = k + 1; i < r; ++i) W[i + (size_t)j * ld] -= W[i + (size_t)k * ld] * wkj;
    }
  }
  return 0;
}
static inline void lu_solve(const real *W, int r, int ld, const int *ipiv, real *b) {
  for (int k = 0; k < r; ++k)
    if (ipiv[k] != k) std::swap(b[k], b[ipiv[k]]);
  for (int k = 0; k < r; ++k)
    for (int i = k + 1; i < r; ++i) b[i] -= W[i + (size_t)k * ld] * b[k];
  for (int k = r - 1; k >= 0; --k) {
    b[k] /= W[k + (size_t)k * ld];
    for (int i = 0; i < k; ++i) b[i] -= W[i + (size_t)k * ld] * b[k];
  }
}
static inline void upper_solve(const real *R, long ld, int r, real *b) { // R u = b
  for (int k = r - 1; k >= 0; --k) {
    b[k] /= R[k + (size_t)k * ld];
    for (int i = 0; i < k; ++i) b[i] -= R[i + (size_t)k * ld] * b[k];
  }
}

} // namespace scsamd
