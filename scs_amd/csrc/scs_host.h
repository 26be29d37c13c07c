// scs_host.h -- host-side pieces shared by the B2 driver (setup-time only).
#pragma once
#include "common.h"

namespace scsamd {

typedef scs_float real;

// Internal CSC view: 32-bit indices whatever the ABI's scs_int is (the device kernels index with int; -DDLONG only
// widens the boundary structs).  Without DLONG it aliases the caller's arrays, with DLONG CscArg narrows a copy.
struct CscView {
  real *x;
  int *i;  // row indices: 32-bit
  eoff *p; // column pointers: entry positions
  int m, n;
};

// owned host CSC copy (deep copy of the user's matrix; equilibrated in place)
struct HostCsc {
  int m = 0, n = 0;
  std::vector<eoff> p;
  std::vector<int> i;
  std::vector<real> x;
  void copy_from(const ScsMatrix *M) {
    m = (int)M->m;
    n = (int)M->n;
    const size_t nnz = (size_t)M->p[M->n];
    p.resize((size_t)n + 1);
    i.resize(nnz);
    for (size_t j = 0; j <= (size_t)n; ++j) p[j] = (eoff)M->p[j];
    for (size_t j = 0; j < nnz; ++j) i[j] = (int)M->i[j];
    x.assign(M->x, M->x + nnz);
  }
  CscView view() { return CscView{x.data(), i.data(), p.data(), m, n}; }
};

// a caller's ScsMatrix as a CscView: aliased when scs_int is int, narrowed into an owned copy under -DDLONG
struct CscArg {
  HostCsc own;
  CscView v{nullptr, nullptr, nullptr, 0, 0};
  bool present = false;
  explicit CscArg(const ScsMatrix *M) {
    if (!M) return;
    present = true;
    if (sizeof(scs_int) == sizeof(int)) {
      v = CscView{M->x, reinterpret_cast<int *>(M->i), reinterpret_cast<eoff *>(M->p), (int)M->m, (int)M->n};
    } else {
      own.copy_from(M);
      v = own.view();
    }
  }
  const CscView *ptr() const { return present ? &v : nullptr; }
};

// every size and index of a caller's matrix must fit the device's indexing: m, n and every row index 32-bit; the number of nonzeros
// 32-bit too unless entry positions are 64-bit (eoff: the DLONG build)
inline bool fits_int32(const ScsMatrix *M) {
  if (!M || !M->p) return true; // missing arrays are validate_csc's business
  if ((long long)M->m > 2147483647LL || (long long)M->n > 2147483647LL || M->n < 0) return false;
  return sizeof(eoff) == 8 || (long long)M->p[M->n] < 2147483647LL;
}

// reference include/scs_work.h:24-29 (ScsScaling)
struct Scaling {
  std::vector<real> D, E;
  real primal_scale = 1, dual_scale = 1;
};

// pattern of the CSR copy of a CSC matrix: row pointers, column indices and, for every
// CSR entry, its position in the CSC arrays (so values can be gathered, never re-sorted)
// (round 5) `dev`: both orientations of the equilibrated matrix as equilibrate_dev left them in HBM -- CSC = CSR(A') in cp / ci / cx,
// CSR(A) in rp / rj / rx -- so that LinSys::init adopts them instead of gathering the values on the host and uploading 2 x 12 B / nnz
// again.  With `dev.valid` the host copies rj / pos may be empty (the transpose was built on the device): they are fetched on demand.
struct DevMatrices {
  DevBuf<eoff> cp, rp, rpos;
  DevBuf<int> ci, rj;
  DevBuf<real> cx, rx;
  bool valid = false;
};
struct CsrPattern {
  std::vector<eoff> rp, pos; // row pointers; CSC position of every CSR entry
  std::vector<int> rj;
  DevMatrices dev;
  bool built_on_device = false; // the transpose (rj, pos) was formed on the device: the host copies are empty
  bool empty() const { return rp.empty(); }
  void clear() {
    rp = std::vector<eoff>();
    rj = std::vector<int>();
    pos = std::vector<eoff>();
    for (DevBuf<eoff> *b : {&dev.cp, &dev.rp, &dev.rpos}) b->release();
    dev.ci.release();
    dev.rj.release();
    dev.cx.release();
    dev.rx.release();
    dev.valid = false;
    built_on_device = false;
  }
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
