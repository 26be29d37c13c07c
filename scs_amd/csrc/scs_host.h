// scs_host.h -- host-side pieces shared by the B2 driver (setup-time only).
#pragma once
#include "common.h"

namespace scsamd {

typedef scs_float real;

// owned host CSC copy (deep copy of the user's matrix; equilibrated in place)
struct HostCsc {
  int m = 0, n = 0;
  std::vector<int> p, i;
  std::vector<real> x;
  void copy_from(const ScsMatrix *M) {
    m = M->m;
    n = M->n;
    const size_t nnz = (size_t)M->p[M->n];
    p.assign(M->p, M->p + M->n + 1);
    i.assign(M->i, M->i + nnz);
    x.assign(M->x, M->x + nnz);
  }
  ScsMatrix view() { return ScsMatrix{x.data(), i.data(), p.data(), m, n}; }
};

// reference include/scs_work.h:24-29 (ScsScaling)
struct Scaling {
  std::vector<real> D, E;
  real primal_scale = 1, dual_scale = 1;
};

// pattern of the CSR copy of a CSC matrix: row pointers, column indices and, for every
// CSR entry, its position in the CSC arrays (so values can be gathered, never re-sorted)
struct CsrPattern {
  std::vector<int> rp, rj, pos;
  bool empty() const { return rp.empty(); }
};

std::vector<int> cone_segments(const ScsCone *k);
void equilibrate(HostCsc *P, HostCsc &A, const ScsCone *k, Scaling &sc);
// the same passes on the device (equilibrate_dev.hip); bit-identical results
void equilibrate_dev(HostCsc *P, HostCsc &A, const ScsCone *k, Scaling &sc, hipStream_t st,
                     CsrPattern *csr_cache);
void normalize_b_c(Scaling &sc, real *b, real *c);
void normalize_sol(const Scaling &sc, real *x, real *y, real *s);
void un_normalize_sol(const Scaling &sc, real *x, real *y, real *s);
int validate_csc(const ScsMatrix *M, int rows, int cols, bool upper_only, const char *name);

int write_problem(const ScsData *d, const ScsCone *k, const ScsSettings *s, const char *filename);
int read_problem(const char *filename, ScsData **d, ScsCone **k, ScsSettings **s);
void free_problem(ScsData *d, ScsCone *k, ScsSettings *s);

// ---- host Anderson acceleration (reference include/aa.h:66-143) --------------
struct AaHost;
AaHost *aa_host_init(int dim, int mem, int min_len, int type1, real regularization, real relaxation,
                     real safeguard_factor, real max_weight_norm, int ir_max_steps);
real aa_host_apply(real *f, const real *x, AaHost *a);
int aa_host_safeguard(real *f_new, real *x_new, AaHost *a);
void aa_host_reset(AaHost *a);
void aa_host_finish(AaHost *a);
void aa_host_stats(const AaHost *a, AaStats *out);

// ---- device Anderson acceleration (aa_dev.hip): same contract, f and x are device
// pointers and all O(dim) work stays in HBM
struct AaDev;
AaDev *aa_dev_init(int dim, int mem, int min_len, int type1, real regularization, real relaxation,
                   real safeguard_factor, real max_weight_norm, int ir_max_steps, hipStream_t st);
real aa_dev_apply(real *f, const real *x, AaDev *a);
int aa_dev_safeguard(real *f_new, real *x_new, AaDev *a);
void aa_dev_reset(AaDev *a);
void aa_dev_finish(AaDev *a);
void aa_dev_stats(const AaDev *a, AaStats *out);

} // namespace scsamd
