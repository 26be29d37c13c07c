// aa_dev.hip -- device-resident Anderson acceleration of the ADMM fixed-point map
// (SURVEY.md 8f item 1).  Same algorithm and the same host-side decisions as the
// host version (aa_host.cpp, restating reference src/aa.c:236-967), but every
// O(dim) operation runs on the GPU and v / v_prev never leave HBM:
//
//   update_accel_params (aa.c:340-391)   one fused kernel: S, D, Y columns, g, the
//                                        x/f/g_prev copies and three sums of squares
//   solve (aa.c:422-655)                 column-pivoted Householder QR of [A; sqrt(r) I]
//                                        as a left-looking sweep over a tall-skinny
//                                        panel that ALSO carries [Y; sqrt(r) I] (type-I)
//                                        and [g; 0]: reflector k is applied to every
//                                        remaining column in one "dots" pass and one
//                                        "update" pass, so Q'[g;0] and Q'[Y;..] fall out
//                                        of the factorisation instead of being separate
//                                        ormqr sweeps.  Columns are never swapped in
//                                        memory (pivoting permutes indices), the
//                                        reflectors are never stored (nothing reads
//                                        them afterwards).
//   f -= D gamma, relaxation (:393-410)  one kernel
//   safeguard (aa.c:856-899)             one reduction, two D2D copies on rejection
//
// The host keeps what is O(mem^2): pivot choice with LAPACK-style norm downdating,
// reflector scalars, rank truncation, the mem x mem solve with iterative refinement
// (aa_small.h).  One small read-back per reflector drives those decisions.
// Reductions are two-level with fixed order (deterministic run to run); they are not
// the host's summation order, so device and host AA agree to rounding, not bitwise.
#include "aa_small.h"
#include <cfloat>

namespace scsamd {

constexpr int AA_BATCH = 16; // columns carried per launch
constexpr int AA_GRID = 512; // workgroups per tall-skinny pass (= partials per column)

struct ColSet {
  int n;
  int col[AA_BATCH];
};

// ---- kernels ---------------------------------------------------------------------
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_aa_seed(const real *__restrict__ x, const real *__restrict__ f, real *__restrict__ ax, real *__restrict__ af,
          real *__restrict__ g_prev, long dim) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (long)gridDim.x * blockDim.x) {
    const real xi = x[i], fi = f[i];
    ax[i] = xi;
    af[i] = fi;
    g_prev[i] = xi - fi;
  }
}

// part: [3][gridDim.x] sums of squares of the new S column, Y column, g
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_aa_update(const real *__restrict__ x, const real *__restrict__ f, real *__restrict__ ax, real *__restrict__ af,
            real *__restrict__ g, real *__restrict__ g_prev, real *__restrict__ sc, real *__restrict__ dc,
            real *__restrict__ yc, real *__restrict__ x_work, long dim, real *__restrict__ part) {
  __shared__ real sh[SCSAMD_BLOCK / 64];
  real ss = 0, sy = 0, sg = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (long)gridDim.x * blockDim.x) {
    const real xi = x[i], fi = f[i];
    const real s = xi - ax[i], d = fi - af[i], gi = xi - fi, y = gi - g_prev[i];
    sc[i] = s;
    dc[i] = d;
    yc[i] = y;
    g[i] = gi;
    g_prev[i] = gi;
    ax[i] = xi;
    af[i] = fi;
    if (x_work) x_work[i] = xi;
    ss += s * s;
    sy += y * y;
    sg += gi * gi;
  }
  ss = block_sum(ss, sh);
  sy = block_sum(sy, sh);
  sg = block_sum(sg, sh);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = ss;
    part[gridDim.x + blockIdx.x] = sy;
    part[2 * gridDim.x + blockIdx.x] = sg;
  }
}

// Panel columns: [0, mem) = [A_j; sqrt(r) e_j], [mem, 2 mem) = [Y_j; sqrt(r) e_j] (type-I
// only), last = [g; 0].  blockIdx.y picks the column.
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_aa_build(real *__restrict__ Q, long ld, long dim, long aug, int mem, int len, int type1,
           const real *__restrict__ A_src, const real *__restrict__ Y, const real *__restrict__ g, real sqrt_r) {
  const int cy = blockIdx.y; // 0..len-1: A, len..2len-1: B (type1), last: c
  const real *src;
  int phys, unit;
  if (cy < len) {
    src = A_src + (size_t)cy * dim;
    phys = cy;
    unit = cy;
  } else if (type1 && cy < 2 * len) {
    src = Y + (size_t)(cy - len) * dim;
    phys = mem + (cy - len);
    unit = cy - len;
  } else {
    src = g;
    phys = type1 ? 2 * mem : mem;
    unit = -1;
  }
  real *dst = Q + (size_t)phys * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < aug; i += (long)gridDim.x * blockDim.x) {
    real v;
    if (i < dim) v = src[i];
    else v = (unit >= 0 && i - dim == unit) ? sqrt_r : (real)0;
    dst[i] = v;
  }
}

// part[c][wg] = sum_{i >= lo} Q[piv][i] * Q[col_c][i]
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_qr_dots(const real *__restrict__ Q, long ld, long aug, long lo, int piv, ColSet cs, real *__restrict__ part) {
  __shared__ real sh[SCSAMD_BLOCK / 64][AA_BATCH];
  real acc[AA_BATCH];
#pragma unroll
  for (int c = 0; c < AA_BATCH; ++c) acc[c] = 0;
  const real *vp = Q + (size_t)piv * ld;
  for (long i = lo + (long)blockIdx.x * blockDim.x + threadIdx.x; i < aug; i += (long)gridDim.x * blockDim.x) {
    const real v = vp[i];
#pragma unroll
    for (int c = 0; c < AA_BATCH; ++c)
      if (c < cs.n) acc[c] += v * Q[(size_t)cs.col[c] * ld + i];
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < AA_BATCH; ++c) {
    if (c < cs.n) {
      const real s = wave_sum(acc[c]);
      if (l == 0) sh[w][c] = s;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < cs.n) {
    real s = sh[0][threadIdx.x];
    for (int k = 1; k < SCSAMD_BLOCK / 64; ++k) s += sh[k][threadIdx.x];
    part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
  }
}

// one workgroup: w_c = tau (Q[col_c][k] + vscale * dot_c); Q[col_c][k] -= w_c; the new
// row-k entries go to info_ck (pivot norm downdating); R_kk = beta into the pivot column.
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_qr_w(real *__restrict__ Q, long ld, long k, int piv, real tau, real vscale, real beta, int set_beta, ColSet cs,
       const real *__restrict__ part, int nparts, real *__restrict__ W, real *__restrict__ info_ck) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int c = w; c < cs.n; c += SCSAMD_BLOCK / 64) {
    real s = 0;
    for (int i = l; i < nparts; i += 64) s += part[(size_t)c * nparts + i];
    s = wave_sum(s);
    if (l == 0) {
      real *e = Q + (size_t)cs.col[c] * ld + k;
      const real ck = *e;
      const real wc = tau * (ck + vscale * s);
      *e = ck - wc;
      W[c] = wc;
      info_ck[c] = ck - wc;
    }
  }
  if (set_beta && threadIdx.x == 0) Q[(size_t)piv * ld + k] = beta;
}

// rows i >= lo of every column in the set: x -= W_c * vscale * Q[piv][i] (if do_update);
// statistics for the next pivot step: elem[c] = x at row lo, part_ss[c][wg] = sum of
// squares over rows > lo.
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_qr_update(real *__restrict__ Q, long ld, long aug, long lo, int piv, real vscale, int do_update, ColSet cs,
            const real *__restrict__ W, real *__restrict__ part_ss, real *__restrict__ elem) {
  __shared__ real sh[SCSAMD_BLOCK / 64][AA_BATCH];
  real ss[AA_BATCH], wv[AA_BATCH];
#pragma unroll
  for (int c = 0; c < AA_BATCH; ++c) {
    ss[c] = 0;
    wv[c] = (do_update && c < cs.n) ? W[c] : (real)0;
  }
  const real *vp = Q + (size_t)piv * ld;
  for (long i = lo + (long)blockIdx.x * blockDim.x + threadIdx.x; i < aug; i += (long)gridDim.x * blockDim.x) {
    const real v = do_update ? vscale * vp[i] : (real)0;
#pragma unroll
    for (int c = 0; c < AA_BATCH; ++c) {
      if (c < cs.n) {
        real *e = Q + (size_t)cs.col[c] * ld + i;
        real x = *e;
        if (do_update) {
          x -= wv[c] * v;
          *e = x;
        }
        if (i > lo) ss[c] += x * x;
        else elem[c] = x;
      }
    }
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < AA_BATCH; ++c) {
    if (c < cs.n) {
      const real s = wave_sum(ss[c]);
      if (l == 0) sh[w][c] = s;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < cs.n) {
    real s = sh[0][threadIdx.x];
    for (int k = 1; k < SCSAMD_BLOCK / 64; ++k) s += sh[k][threadIdx.x];
    part_ss[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
  }
}

// f -= D gamma; with relaxation: x_work -= S gamma, f = relax f + (1-relax) x_work
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_aa_combine(real *__restrict__ f, const real *__restrict__ D, const real *__restrict__ S,
             real *__restrict__ x_work, const real *__restrict__ gamma, int len, long dim, real relaxation) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (long)gridDim.x * blockDim.x) {
    real fi = f[i];
    for (int j = 0; j < len; ++j) {
      const real gj = gamma[j];
      if (gj != 0) fi -= D[(size_t)j * dim + i] * gj;
    }
    if (x_work) {
      real xw = x_work[i];
      for (int j = 0; j < len; ++j) {
        const real gj = gamma[j];
        if (gj != 0) xw -= S[(size_t)j * dim + i] * gj;
      }
      x_work[i] = xw;
      fi = relaxation * fi + ((real)1. - relaxation) * xw;
    }
    f[i] = fi;
  }
}

__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_aa_diff_sumsq(const real *__restrict__ a, const real *__restrict__ b, long dim, real *__restrict__ part) {
  __shared__ real sh[SCSAMD_BLOCK / 64];
  real s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += (long)gridDim.x * blockDim.x) {
    const real d = a[i] - b[i];
    s += d * d;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---- state -------------------------------------------------------------------------
struct AaDev {
  int type1 = 1, mem = 0, min_len = 0, iter = 0, success = 0, ir_max_steps = 0;
  long dim = 0, aug = 0, ld = 0;
  real relaxation = 1, regularization = 0, safeguard_factor = 1, max_weight_norm = 0;
  real norm_g = 0;
  hipStream_t st = nullptr;
  int grid = 1;
  DevBuf<real> x, f, g, g_prev, x_work, Y, S, D, Q;
  DevBuf<real> part;  // [2 * max_batches][AA_BATCH][grid] partial sums
  DevBuf<real> small; // per batch: W[AA_BATCH] | ck[AA_BATCH] | elem[AA_BATCH]; then gamma[mem]
  PinnedBuf<real> h;  // read-back area
  int max_batches = 1;
  std::vector<real> nrm_s_col, nrm_y_col, cn, cn0, tau, E, SS, CK, top, Rm, W, W_orig, gamma, gamma_red, c_top,
      ir_res;
  std::vector<int> jpvt, ipiv;
  AaStats stt;
  // offsets inside `small`
  real *Wd(int b) { return small.p + (size_t)b * 3 * AA_BATCH; }
  real *ckd(int b) { return Wd(b) + AA_BATCH; }
  real *elemd(int b) { return Wd(b) + 2 * AA_BATCH; }
  real *gammad() { return small.p + (size_t)max_batches * 3 * AA_BATCH; }
  real *part_dot(int b) { return part.p + (size_t)b * AA_BATCH * grid; }
  real *part_ss(int b) { return part.p + (size_t)(max_batches + b) * AA_BATCH * grid; }
  int col_c() const { return type1 ? 2 * mem : mem; }
};

AaDev *aa_dev_init(int dim, int mem, int min_len, int type1, real regularization, real relaxation,
                   real safeguard_factor, real max_weight_norm, int ir_max_steps, hipStream_t st) {
  const int memc = std::min(mem, dim);
  if (dim <= 0 || mem < 0 || !std::isfinite((double)regularization) || relaxation < 0 || relaxation > 2 ||
      safeguard_factor < 0 || max_weight_norm <= 0 || ir_max_steps < 0 || (memc > 0 && min_len < 1)) {
    printf("Invalid AA parameters.\n");
    return nullptr;
  }
  AaDev *a = new AaDev();
  a->type1 = type1;
  a->dim = dim;
  a->mem = memc;
  a->min_len = memc > 0 ? std::min(min_len, memc) : 0;
  a->regularization = regularization;
  a->relaxation = relaxation;
  a->safeguard_factor = safeguard_factor;
  a->max_weight_norm = max_weight_norm;
  a->ir_max_steps = ir_max_steps;
  a->st = st;
  memset(&a->stt, 0, sizeof a->stt);
  a->stt.last_aa_norm = (real)NAN;
  if (memc <= 0) return a;
  try {
    const size_t d = (size_t)dim, m = (size_t)memc;
    a->aug = dim + memc;
    a->ld = (a->aug + 7) & ~7L;
    a->grid = std::max(1, std::min(AA_GRID, ceil_div(a->aug, SCSAMD_BLOCK)));
    const int ncols = (type1 ? 2 : 1) * memc + 1;
    a->max_batches = ceil_div(ncols, AA_BATCH);
    a->x.alloc(d); a->f.alloc(d); a->g.alloc(d); a->g_prev.alloc(d);
    if (relaxation != (real)1.0) a->x_work.alloc(d);
    a->Y.alloc(d * m); a->S.alloc(d * m); a->D.alloc(d * m);
    a->Q.alloc((size_t)a->ld * ncols);
    a->part.alloc((size_t)2 * a->max_batches * AA_BATCH * a->grid + 3 * a->grid);
    a->small.alloc((size_t)a->max_batches * 3 * AA_BATCH + m);
    a->h.alloc(std::max((size_t)a->max_batches * (3 * AA_BATCH + (size_t)AA_BATCH * a->grid) + 3 * a->grid,
                        (size_t)ncols * m));
    a->nrm_s_col.assign(m, 0); a->nrm_y_col.assign(m, 0);
    a->cn.assign(m, 0); a->cn0.assign(m, 0); a->tau.assign(m, 0);
    a->E.assign(ncols, 0); a->SS.assign(ncols, 0); a->CK.assign(ncols, 0);
    a->top.assign((size_t)ncols * m, 0); a->Rm.assign(m * m, 0);
    a->W.assign(m * m, 0); a->W_orig.assign(m * m, 0);
    a->gamma.assign(m, 0); a->gamma_red.assign(m, 0); a->c_top.assign(m, 0); a->ir_res.assign(m, 0);
    a->jpvt.assign(m, 0); a->ipiv.assign(m, 0);
  } catch (const std::exception &e) {
    printf("Failed to allocate memory for AA (%s).\n", e.what());
    delete a;
    return nullptr;
  }
  return a;
}

void aa_dev_reset(AaDev *a) { // aa.c:934-967
  if (!a) return;
  a->iter = 0;
  a->success = 0;
  a->norm_g = 0;
  std::fill(a->nrm_s_col.begin(), a->nrm_s_col.end(), (real)0);
  std::fill(a->nrm_y_col.begin(), a->nrm_y_col.end(), (real)0);
}
void aa_dev_finish(AaDev *a) { delete a; }
void aa_dev_stats(const AaDev *a, AaStats *out) {
  *out = a->stt;
  out->iter = a->iter;
}

// ---- the panel sweep ---------------------------------------------------------------
static std::vector<ColSet> make_batches(const std::vector<int> &cols) {
  std::vector<ColSet> out;
  for (size_t i = 0; i < cols.size(); i += AA_BATCH) {
    ColSet cs;
    cs.n = (int)std::min((size_t)AA_BATCH, cols.size() - i);
    for (int c = 0; c < AA_BATCH; ++c) cs.col[c] = c < cs.n ? cols[i + c] : 0;
    out.push_back(cs);
  }
  return out;
}

// Statistics pass and/or reflector application over `cols`; afterwards E/SS (and CK when
// a reflector was applied) hold, per physical column, the row-lo entry, the sum of
// squares below it and the updated row-k entry.  `n_stat` = how many leading entries of
// `cols` the host needs back (the pivot candidates).
static void sweep(AaDev *a, const std::vector<int> &cols, int n_stat, long k, int piv, bool apply, real tau,
                  real vscale, real beta) {
  const std::vector<ColSet> bs = make_batches(cols);
  const long lo = k + 1;
  const int G = a->grid;
  for (size_t b = 0; b < bs.size(); ++b) {
    if (apply) {
      hipLaunchKernelGGL(k_qr_dots, dim3(G), dim3(SCSAMD_BLOCK), 0, a->st, a->Q.p, a->ld, a->aug, lo, piv, bs[b],
                         a->part_dot((int)b));
      hipLaunchKernelGGL(k_qr_w, dim3(1), dim3(SCSAMD_BLOCK), 0, a->st, a->Q.p, a->ld, k, piv, tau, vscale, beta,
                         b == 0 ? 1 : 0, bs[b], a->part_dot((int)b), G, a->Wd((int)b), a->ckd((int)b));
    }
    const bool need_stats = (int)(b * AA_BATCH) < n_stat;
    if (apply || need_stats)
      hipLaunchKernelGGL(k_qr_update, dim3(G), dim3(SCSAMD_BLOCK), 0, a->st, a->Q.p, a->ld, a->aug, lo, piv, vscale,
                         apply ? 1 : 0, bs[b], a->Wd((int)b), a->part_ss((int)b), a->elemd((int)b));
  }
  HIP_CHECK(hipGetLastError());
  if (n_stat <= 0) return;
  const int nb = ceil_div(n_stat, AA_BATCH);
  real *h_small = a->h.p, *h_ss = a->h.p + (size_t)a->max_batches * 3 * AA_BATCH;
  HIP_CHECK(hipMemcpyAsync(h_small, a->small.p, (size_t)nb * 3 * AA_BATCH * sizeof(real), hipMemcpyDeviceToHost,
                           a->st));
  HIP_CHECK(hipMemcpyAsync(h_ss, a->part_ss(0), (size_t)nb * AA_BATCH * G * sizeof(real), hipMemcpyDeviceToHost,
                           a->st));
  HIP_CHECK(hipStreamSynchronize(a->st));
  for (int i = 0; i < n_stat; ++i) {
    const int b = i / AA_BATCH, c = i % AA_BATCH, col = cols[i];
    const real *sm = h_small + (size_t)b * 3 * AA_BATCH;
    if (apply) a->CK[col] = sm[AA_BATCH + c];
    else a->CK[col] = a->E[col]; // row k untouched: it is the row-lo entry of the previous pass
    real s = 0;
    const real *ps = h_ss + ((size_t)b * AA_BATCH + c) * G;
    for (int w = 0; w < G; ++w) s += ps[w];
    a->E[col] = sm[2 * AA_BATCH + c];
    a->SS[col] = s;
  }
}

static real aa_dev_solve(real *f, AaDev *a, int len) {
  const long dim = a->dim;
  const int mem = a->mem;
  const real *A_src = a->type1 ? a->S.p : a->Y.p;
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
  const int ncarry = (a->type1 ? len : 0) + 1;
  hipLaunchKernelGGL(k_aa_build, dim3(a->grid, len + ncarry), dim3(SCSAMD_BLOCK), 0, a->st, a->Q.p, a->ld, dim,
                     a->aug, mem, len, a->type1, A_src, a->Y.p, a->g.p, sqrt_r);
  // carried (never pivoted) columns: B then c
  std::vector<int> carried;
  if (a->type1)
    for (int j = 0; j < len; ++j) carried.push_back(mem + j);
  carried.push_back(a->col_c());
  // initial statistics of the pivot candidates
  std::vector<int> cols;
  for (int j = 0; j < len; ++j) cols.push_back(j);
  sweep(a, cols, len, -1, 0, false, 0, 0, 0);
  for (int j = 0; j < len; ++j) {
    a->jpvt[j] = j;
    a->cn[j] = a->cn0[j] = std::sqrt(a->E[j] * a->E[j] + a->SS[j]);
  }
  const real tol3z = std::sqrt((real)(sizeof(real) == 8 ? DBL_EPSILON : FLT_EPSILON));
  for (int k = 0; k < len; ++k) {
    int piv = k;
    for (int j = k + 1; j < len; ++j)
      if (a->cn[j] > a->cn[piv]) piv = j;
    if (piv != k) {
      std::swap(a->jpvt[k], a->jpvt[piv]);
      a->cn[piv] = a->cn[k];
      a->cn0[piv] = a->cn0[k];
    }
    const int P = a->jpvt[k];
    const real alpha = a->E[P], xnorm = std::sqrt(a->SS[P]);
    real beta = alpha, sc = 0;
    if (xnorm == 0) {
      a->tau[k] = 0;
    } else {
      beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
      a->tau[k] = (beta - alpha) / beta;
      sc = (real)1 / (alpha - beta);
    }
    cols.clear();
    for (int j = k + 1; j < len; ++j) cols.push_back(a->jpvt[j]);
    const int n_stat = (int)cols.size();
    for (int c : carried) cols.push_back(c);
    sweep(a, cols, n_stat, k, P, a->tau[k] != 0, a->tau[k], sc, beta);
    for (int j = k + 1; j < len; ++j) { // norm downdating, as dgeqp3 / aa_host.cpp
      const int col = a->jpvt[j];
      if (a->cn[j] != 0) {
        real t = std::fabs(a->CK[col]) / a->cn[j];
        t = std::max((real)0, (1 + t) * (1 - t));
        const real t2 = t * (a->cn[j] / a->cn0[j]) * (a->cn[j] / a->cn0[j]);
        if (t2 <= tol3z) {
          a->cn[j] = std::sqrt(a->E[col] * a->E[col] + a->SS[col]);
          a->cn0[j] = a->cn[j];
        } else {
          a->cn[j] *= std::sqrt(t);
        }
      }
    }
  }
  // top len rows of every panel column: R, W = top of Q'[Y_piv; ..], c_top
  const int ncols = (a->type1 ? 2 : 1) * mem + 1;
  HIP_CHECK(hipMemcpy2DAsync(a->h.p, (size_t)mem * sizeof(real), a->Q.p, (size_t)a->ld * sizeof(real),
                             (size_t)len * sizeof(real), ncols, hipMemcpyDeviceToHost, a->st));
  HIP_CHECK(hipStreamSynchronize(a->st));
  memcpy(a->top.data(), a->h.p, (size_t)ncols * mem * sizeof(real));
  auto topv = [&](int col, int row) -> real { return a->top[(size_t)col * mem + row]; };
  int rank = 0, info = 0;
  {
    const real r11 = std::fabs(topv(a->jpvt[0], 0));
    if (r11 > 0) {
      const real tol = r11 * (real)len * (real)(sizeof(real) == 8 ? DBL_EPSILON : FLT_EPSILON);
      for (rank = 0; rank < len; ++rank)
        if (std::fabs(topv(a->jpvt[rank], rank)) < tol) break;
    }
    if (rank == 0) info = 1;
  }
  if (info == 0) {
    for (int i = 0; i < rank; ++i) a->c_top[i] = topv(a->col_c(), i);
    if (a->type1) {
      for (int i = 0; i < rank; ++i)
        for (int rr = 0; rr < rank; ++rr) {
          a->W[(size_t)i * mem + rr] = topv(mem + a->jpvt[i], rr);
          a->W_orig[(size_t)i * mem + rr] = a->W[(size_t)i * mem + rr];
        }
      memcpy(a->gamma_red.data(), a->c_top.data(), rank * sizeof(real));
      info = lu_factor(a->W.data(), rank, mem, a->ipiv.data());
      if (info == 0) {
        lu_solve(a->W.data(), rank, mem, a->ipiv.data(), a->gamma_red.data());
        real prev = 0;
        for (int k = 0; k < a->ir_max_steps; ++k) { // aa.c:530-552
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
      for (int j = 0; j < rank; ++j)
        for (int i = 0; i <= j; ++i) a->Rm[i + (size_t)j * mem] = topv(a->jpvt[j], i);
      memcpy(a->gamma_red.data(), a->c_top.data(), rank * sizeof(real));
      upper_solve(a->Rm.data(), mem, rank, a->gamma_red.data());
      real prev = 0;
      for (int k = 0; k < a->ir_max_steps; ++k) { // aa.c:566-585
        for (int i = 0; i < rank; ++i) {
          real s = 0;
          for (int j = i; j < rank; ++j) s += a->Rm[i + (size_t)j * mem] * a->gamma_red[j];
          a->ir_res[i] = a->c_top[i] - s;
        }
        upper_solve(a->Rm.data(), mem, rank, a->ir_res.data());
        const real dn = nrm2(a->ir_res.data(), rank);
        for (int i = 0; i < rank; ++i) a->gamma_red[i] += a->ir_res[i];
        if (k > 0 && dn >= (real)0.5 * prev) break;
        prev = dn;
      }
    }
    if (info == 0) {
      for (int i = 0; i < len; ++i) a->gamma[i] = 0;
      for (int i = 0; i < rank; ++i) a->gamma[a->jpvt[i]] = a->gamma_red[i];
    }
  }
  real aa_norm = info == 0 ? nrm2(a->gamma.data(), len) : (real)-1.0;
  a->stt.last_rank = rank;
  a->stt.last_regularization = r;
  a->stt.last_aa_norm = (info == 0 && std::isfinite((double)aa_norm)) ? aa_norm : (real)NAN;
  if (info != 0 || !std::isfinite((double)aa_norm) || aa_norm >= a->max_weight_norm) {
    if (rank == 0) a->stt.n_reject_rank0++;
    else if (info != 0) a->stt.n_reject_lapack++;
    else if (!std::isfinite((double)aa_norm)) a->stt.n_reject_nonfinite++;
    else a->stt.n_reject_weight_cap++;
    a->success = 0;
    aa_dev_reset(a);
    if (!std::isfinite((double)aa_norm)) aa_norm = -1.0;
    return aa_norm < 0 ? aa_norm : -aa_norm;
  }
  HIP_CHECK(hipMemcpyAsync(a->gammad(), a->gamma.data(), (size_t)len * sizeof(real), hipMemcpyHostToDevice, a->st));
  const int g1 = std::max(1, std::min(4096, ceil_div(dim, SCSAMD_BLOCK)));
  hipLaunchKernelGGL(k_aa_combine, dim3(g1), dim3(SCSAMD_BLOCK), 0, a->st, f, a->D.p, a->S.p,
                     a->x_work.p /* null unless relaxation != 1 */, a->gammad(), len, dim, a->relaxation);
  HIP_CHECK(hipGetLastError());
  // gamma lives in pageable host memory: make sure the copy has consumed it
  HIP_CHECK(hipStreamSynchronize(a->st));
  a->success = 1;
  return aa_norm;
}

// f, x: device pointers (length dim) on the AA stream
real aa_dev_apply(real *f, const real *x, AaDev *a) { // aa.c:822-854
  real aa_norm = 0;
  const int len = std::min(a->iter, a->mem);
  const long dim = a->dim;
  a->success = 0;
  if (a->mem <= 0) return aa_norm;
  const int g1 = std::max(1, std::min(4096, ceil_div(dim, SCSAMD_BLOCK)));
  if (a->iter == 0) { // init_accel_params, aa.c:293-307
    hipLaunchKernelGGL(k_aa_seed, dim3(g1), dim3(SCSAMD_BLOCK), 0, a->st, x, f, a->x.p, a->f.p, a->g_prev.p, dim);
    HIP_CHECK(hipGetLastError());
    a->iter++;
    return aa_norm;
  }
  { // update_accel_params, aa.c:340-391
    const int idx = (a->iter - 1) % a->mem;
    const int G = a->grid;
    real *p3 = a->part.p + (size_t)2 * a->max_batches * AA_BATCH * G;
    hipLaunchKernelGGL(k_aa_update, dim3(G), dim3(SCSAMD_BLOCK), 0, a->st, x, f, a->x.p, a->f.p, a->g.p,
                       a->g_prev.p, a->S.p + (size_t)idx * dim, a->D.p + (size_t)idx * dim,
                       a->Y.p + (size_t)idx * dim, a->x_work.p, dim, p3);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(a->h.p, p3, (size_t)3 * G * sizeof(real), hipMemcpyDeviceToHost, a->st));
    HIP_CHECK(hipStreamSynchronize(a->st));
    real s3[3] = {0, 0, 0};
    for (int q = 0; q < 3; ++q)
      for (int w = 0; w < G; ++w) s3[q] += a->h.p[(size_t)q * G + w];
    a->nrm_s_col[idx] = std::sqrt(s3[0]);
    a->nrm_y_col[idx] = std::sqrt(s3[1]);
    a->norm_g = std::sqrt(s3[2]);
  }
  if (a->iter >= a->min_len) {
    aa_norm = aa_dev_solve(f, a, len);
    if (aa_norm > 0) a->stt.n_accept++;
  }
  a->iter++;
  return aa_norm;
}

int aa_dev_safeguard(real *f_new, real *x_new, AaDev *a) { // aa.c:856-899
  if (a->mem <= 0 || !a->success) return 0;
  a->success = 0;
  const long dim = a->dim;
  const int G = a->grid;
  real *p3 = a->part.p + (size_t)2 * a->max_batches * AA_BATCH * G;
  hipLaunchKernelGGL(k_aa_diff_sumsq, dim3(G), dim3(SCSAMD_BLOCK), 0, a->st, x_new, f_new, dim, p3);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(a->h.p, p3, (size_t)G * sizeof(real), hipMemcpyDeviceToHost, a->st));
  HIP_CHECK(hipStreamSynchronize(a->st));
  real s = 0;
  for (int w = 0; w < G; ++w) s += a->h.p[w];
  const real nd = std::sqrt(s);
  if (nd > a->safeguard_factor * a->norm_g) {
    HIP_CHECK(hipMemcpyAsync(f_new, a->f.p, dim * sizeof(real), hipMemcpyDeviceToDevice, a->st));
    HIP_CHECK(hipMemcpyAsync(x_new, a->x.p, dim * sizeof(real), hipMemcpyDeviceToDevice, a->st));
    a->stt.n_safeguard_reject++;
    aa_dev_reset(a);
    return -1;
  }
  return 0;
}

} // namespace scsamd

// ---- C ABI: the device AA behind host pointers, so a test can pin it against the
// reference's aa_apply on the same sequences (vectors are staged through HBM per call;
// inside scs_solve the iterates are already resident) ---------------------------------
using namespace scsamd;
namespace {
struct AaDevHandle {
  AaDev *a = nullptr;
  hipStream_t st = nullptr;
  DevBuf<real> f, x;
  long dim = 0;
};
} // namespace
extern "C" {
void *scs_amd_aa_dev_init(scs_int dim, scs_int mem, scs_int min_len, scs_int type1, scs_float regularization,
                          scs_float relaxation, scs_float safeguard_factor, scs_float max_weight_norm,
                          scs_int ir_max_steps) {
  AaDevHandle *h = nullptr;
  try {
    h = new AaDevHandle();
    HIP_CHECK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
    h->a = aa_dev_init(dim, mem, min_len, type1, regularization, relaxation, safeguard_factor, max_weight_norm,
                       ir_max_steps, h->st);
    if (!h->a) throw HipError("aa_dev_init failed");
    h->dim = dim;
    h->f.alloc(dim);
    h->x.alloc(dim);
    return h;
  } catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    if (h) {
      if (h->a) aa_dev_finish(h->a);
      if (h->st) (void)hipStreamDestroy(h->st);
      delete h;
    }
    return nullptr;
  }
}
scs_float scs_amd_aa_dev_apply(scs_float *f, const scs_float *x, void *hv) {
  AaDevHandle *h = (AaDevHandle *)hv;
  try {
    h->f.upload(f, h->dim, h->st);
    h->x.upload(x, h->dim, h->st);
    const real nrm = aa_dev_apply(h->f.p, h->x.p, h->a);
    h->f.download(f, h->dim, h->st);
    HIP_CHECK(hipStreamSynchronize(h->st));
    return nrm;
  } catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return (scs_float)NAN;
  }
}
scs_int scs_amd_aa_dev_safeguard(scs_float *f_new, scs_float *x_new, void *hv) {
  AaDevHandle *h = (AaDevHandle *)hv;
  try {
    h->f.upload(f_new, h->dim, h->st);
    h->x.upload(x_new, h->dim, h->st);
    const int rc = aa_dev_safeguard(h->f.p, h->x.p, h->a);
    h->f.download(f_new, h->dim, h->st);
    h->x.download(x_new, h->dim, h->st);
    HIP_CHECK(hipStreamSynchronize(h->st));
    return rc;
  } catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return -2;
  }
}
void scs_amd_aa_dev_reset(void *hv) { aa_dev_reset(((AaDevHandle *)hv)->a); }
void scs_amd_aa_dev_get_stats(const void *hv, AaStats *out) { aa_dev_stats(((const AaDevHandle *)hv)->a, out); }
void scs_amd_aa_dev_finish(void *hv) {
  AaDevHandle *h = (AaDevHandle *)hv;
  if (!h) return;
  if (h->st) (void)hipStreamSynchronize(h->st);
  if (h->a) aa_dev_finish(h->a);
  h->f.release();
  h->x.release();
  if (h->st) (void)hipStreamDestroy(h->st);
  delete h;
}
}
