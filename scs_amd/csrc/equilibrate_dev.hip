// equilibrate_dev.hip -- data equilibration on the device.
//
// Same arithmetic, pass for pass, as the host version (equilibrate.cpp, restating
// reference linsys/scs_matrix.c:236-496: 25 Ruiz passes + 1 L2 pass, per-cone
// aggregation src/cones.c:366-379), laid out for the GPU:
//   * both orientations of A live in HBM (CSC = the user's arrays, CSR = a pattern
//     transpose built once on the host); row statistics run one lane per row over the
//     CSR copy, column statistics one lane per column over the CSC copy, and each
//     rescale updates both copies, so no pass ever scatters;
//   * P (upper triangle) is expanded to the full symmetric CSR so that column j's
//     statistic is a plain row reduction; the upper-triangle values are gathered back
//     at the end;
//   * per-cone max: one workgroup per large cone, one lane per small cone.
// Every floating-point sum runs in the host version's order (row entries by ascending
// column, column entries in storage order, full-P row by ascending column) with
// contraction off, max is order-free, and the one order-sensitive aggregate that has
// no cheap ordered device form (the per-cone MEAN of the single L2 pass, up to ~1e5
// rows summed serially) is done on the host between two kernels -- so the result is
// bit-identical to equilibrate.cpp, which tests/test_equilibrate_gpu.py asserts.
#include "scs_host.h"
#include <algorithm>

#pragma clang fp contract(off)

namespace scsamd {

namespace {

constexpr real MIN_NORM_FACTOR = (real)1e-4, MAX_NORM_FACTOR = (real)1e4;
constexpr int SMALL_CONE = 64;

__device__ __forceinline__ real d_limit_scale(real x) {
  x = x < MIN_NORM_FACTOR ? (real)1.0 : x;
  x = x > MAX_NORM_FACTOR ? MAX_NORM_FACTOR : x;
  return x;
}
__device__ __forceinline__ real d_safe_div_pos(real x, real y) { return y < (real)1e-18 ? x / (real)1e-18 : x / y; }
__device__ __forceinline__ real d_inv_sqrt_scale(real v) { return d_safe_div_pos((real)1, sqrt(d_limit_scale(v))); }

// MODE 0: max |x| per row; MODE 1: sqrt(sum x^2) per row.  One lane per row.
template <int MODE>
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_row_stat(const eoff *__restrict__ rp, const real *__restrict__ rx, int rows, real *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  real acc = 0;
  for (eoff k = rp[i]; k < rp[i + 1]; ++k) {
    const real v = rx[k];
    if (MODE == 0) acc = fmax(acc, fabs(v));
    else acc += v * v;
  }
  out[i] = MODE == 0 ? acc : sqrt(acc);
}

// per-cone max for cones longer than SMALL_CONE rows: one workgroup per cone
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_cone_max_big(const int *__restrict__ off, const int *__restrict__ len, real *__restrict__ v) {
  __shared__ real sh[SCSAMD_BLOCK / 64];
  const int o = off[blockIdx.x], L = len[blockIdx.x];
  real w = 0;
  for (int j = threadIdx.x; j < L; j += blockDim.x) w = fmax(w, fabs(v[o + j]));
  w = block_max(w, sh);
  for (int j = threadIdx.x; j < L; j += blockDim.x) v[o + j] = w;
}
// ... and one lane per cone for the short ones
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_cone_max_small(const int *__restrict__ off, const int *__restrict__ len, int ncones, real *__restrict__ v) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncones) return;
  const int o = off[c], L = len[c];
  real w = 0;
  for (int j = 0; j < L; ++j) w = fmax(w, fabs(v[o + j]));
  for (int j = 0; j < L; ++j) v[o + j] = w;
}

__global__ void __launch_bounds__(SCSAMD_BLOCK) k_inv_sqrt_scale(real *__restrict__ v, int len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) v[i] = d_inv_sqrt_scale(v[i]);
}

// column statistic of A (CSC, one lane per column) merged with the matching row of
// the full symmetric P, then E_j = 1/sqrt(clip(.))
template <int MODE>
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_col_stat(const eoff *__restrict__ cp, const real *__restrict__ cx, const eoff *__restrict__ pp,
           const real *__restrict__ px, int cols, real *__restrict__ Et) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  real e = 0;
  if (pp) {
    for (eoff k = pp[j]; k < pp[j + 1]; ++k) {
      const real v = px[k];
      if (MODE == 0) e = fmax(e, fabs(v));
      else e += v * v;
    }
  }
  real a = 0;
  for (eoff k = cp[j]; k < cp[j + 1]; ++k) {
    const real v = cx[k];
    if (MODE == 0) a = fmax(a, fabs(v));
    else a += v * v;
  }
  if (MODE == 0) e = fmax(e, a);
  else e = sqrt(e + a);
  Et[j] = d_inv_sqrt_scale(e);
}

// x[k] *= outer[major] * inner[minor_index[k]]; written so that the product of the two
// scale factors is formed first, as the host does (A.x[k] *= Dt[i] * Et[j])
__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_rescale(const eoff *__restrict__ ptr, const int *__restrict__ idx, real *__restrict__ x, int majors,
          const real *__restrict__ s_major, const real *__restrict__ s_minor) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= majors) return;
  const real sj = s_major[j];
  for (eoff k = ptr[j]; k < ptr[j + 1]; ++k) {
    const real f = s_minor[idx[k]] * sj;
    x[k] *= f;
  }
}

__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_accumulate(real *__restrict__ acc, const real *__restrict__ t, int len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) acc[i] *= t[i];
}

__global__ void __launch_bounds__(SCSAMD_BLOCK)
k_gather(const real *__restrict__ src, const eoff *__restrict__ map, real *__restrict__ dst, long long len) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < len) dst[k] = src[map[k]];
}

inline int grid_for(long long items) { return std::max(1, ceil_div(items, SCSAMD_BLOCK)); }

// ---- pattern transpose on the device (round 5; the host loop below stays as the general path and the oracle:
// SCS_AMD_TRANSPOSE = dev | host | verify).  Bit-identical to the host's counting sort: row counts by integer atomics (order free),
// prefix sum on the host (the row pointers are needed there anyway), every entry takes SOME slot of its row by an atomic, then each
// row's slots -- which hold CSC positions -- are sorted ascending: CSC position order inside a row IS ascending-column order with
// duplicates in storage order, exactly what the host loop produces.
constexpr int TR_SHORT = 32;      // rows up to this many entries: one lane sorts the row (insertion sort)
constexpr int TR_LONG_MAX = 4096; // longer rows: one workgroup, bitonic sort in LDS; beyond this the host builds the pattern
__device__ __forceinline__ eoff tr_atomic_inc(eoff *p) { // entry positions are non-negative: the unsigned add is the signed one
  if (sizeof(eoff) == 8) return (eoff)atomicAdd(reinterpret_cast<unsigned long long *>(p), 1ull);
  return (eoff)atomicAdd(reinterpret_cast<unsigned int *>(p), 1u);
}
__global__ void __launch_bounds__(SCSAMD_BLOCK) k_tr_count(const int *__restrict__ ci, long long nnz, eoff *cnt) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nnz) tr_atomic_inc(&cnt[ci[q] + 1]);
}
__global__ void __launch_bounds__(SCSAMD_BLOCK) k_tr_scatter(const int *__restrict__ ci, long long nnz, eoff *nxt, eoff *rpos) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nnz) rpos[tr_atomic_inc(&nxt[ci[q]])] = (eoff)q;
}
__global__ void __launch_bounds__(SCSAMD_BLOCK) k_tr_sort_short(const eoff *__restrict__ rp, int rows, eoff *rpos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const eoff a = rp[i];
  const eoff len_all = rp[i + 1] - a;
  if (len_all < 2 || len_all > TR_SHORT) return;
  const int len = (int)len_all;
  for (int u = 1; u < len; ++u) {
    const eoff v = rpos[a + u];
    int w = u - 1;
    while (w >= 0 && rpos[a + w] > v) {
      rpos[a + w + 1] = rpos[a + w];
      --w;
    }
    rpos[a + w + 1] = v;
  }
}
__global__ void __launch_bounds__(SCSAMD_BLOCK) k_tr_sort_long(const eoff *__restrict__ rp, const int *__restrict__ rows_long, eoff *rpos) {
  __shared__ eoff key[TR_LONG_MAX];
  const int i = rows_long[blockIdx.x], tid = threadIdx.x;
  const eoff a = rp[i];
  const int len = (int)(rp[i + 1] - a); // <= TR_LONG_MAX (checked on the host)
  int P2 = 64;
  while (P2 < len) P2 <<= 1;
  const eoff pad = sizeof(eoff) == 8 ? (eoff)0x7fffffffffffffffLL : (eoff)0x7fffffff;
  for (int t = tid; t < P2; t += SCSAMD_BLOCK) key[t] = t < len ? rpos[a + t] : pad;
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = tid; e < P2; e += SCSAMD_BLOCK) {
        const int x = e ^ j;
        if (x > e) {
          const bool asc = (e & k) == 0;
          const eoff va = key[e], vb = key[x];
          if ((va > vb) == asc) {
            key[e] = vb;
            key[x] = va;
          }
        }
      }
      __syncthreads();
    }
  for (int t = tid; t < len; t += SCSAMD_BLOCK) rpos[a + t] = key[t];
}
// the column of every CSC entry, then the CSR column indices through the position map
__global__ void __launch_bounds__(SCSAMD_BLOCK) k_tr_expand_cols(const eoff *__restrict__ cp, int cols, int *colidx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  for (eoff q = cp[j]; q < cp[j + 1]; ++q) colidx[q] = j;
}
__global__ void __launch_bounds__(SCSAMD_BLOCK) k_gather_int(const int *__restrict__ src, const eoff *__restrict__ map, int *__restrict__ dst, long long len) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < len) dst[k] = src[map[k]];
}

} // namespace

// A (and P) are the workspace's host CSC copies: their values are replaced by the
// equilibrated ones, sc.D / sc.E receive the accumulated scalings.  `csr_cache`, when
// given, receives the pattern transpose (row pointers, column indices, CSC position of
// every CSR entry) so the caller does not have to transpose again.
void equilibrate_dev(HostCsc *P, HostCsc &A, const ScsCone *k, Scaling &sc, hipStream_t st,
                     CsrPattern *csr_cache) {
  const int m = A.m, n = A.n;
  const long long nnz = A.p[n];
  static const int NUM_RUIZ_PASSES = 25, NUM_L2_PASSES = 1; // scs_matrix.c:15-16
  sc.D.assign((size_t)m, (real)1);
  sc.E.assign((size_t)n, (real)1);
  sc.primal_scale = sc.dual_scale = 1;

  CsrPattern local;
  CsrPattern &R = csr_cache ? *csr_cache : local;
  // ---- the CSC arrays go up first: the pattern transpose below runs on them
  DevBuf<eoff> cp((size_t)n + 1), rp((size_t)m + 1), rpos((size_t)nnz);
  DevBuf<int> ci((size_t)nnz), rj((size_t)nnz);
  DevBuf<real> cx((size_t)nnz), rx((size_t)nnz), Dt((size_t)m), Et((size_t)n), D((size_t)m), E((size_t)n);
  cp.upload(A.p.data(), (size_t)n + 1, st);
  ci.upload(A.i.data(), (size_t)nnz, st);
  cx.upload(A.x.data(), (size_t)nnz, st);
  // ---- pattern transpose: CSR(A) row pointers, column indices and the CSC position of every CSR entry
  int tr_mode = 1; // device
  if (const char *e = opt_get("transpose")) tr_mode = !strcmp(e, "host") ? 0 : (!strcmp(e, "verify") ? 2 : 1);
  auto host_transpose_pattern = [&](CsrPattern &H) { // counting sort; columns ascending inside each row
    H.rp.assign((size_t)m + 1, 0);
    H.rj.resize((size_t)nnz);
    H.pos.resize((size_t)nnz);
    for (long long q = 0; q < nnz; ++q) H.rp[(size_t)A.i[q] + 1]++;
    for (int i = 0; i < m; ++i) H.rp[i + 1] += H.rp[i];
    std::vector<eoff> nxt(H.rp.begin(), H.rp.end() - 1);
    for (int j = 0; j < n; ++j)
      for (eoff q = A.p[j]; q < A.p[j + 1]; ++q) {
        const eoff t = nxt[A.i[q]]++;
        H.rj[t] = j;
        H.pos[t] = q;
      }
  };
  bool tr_on_dev = false;
  if (tr_mode != 0 && nnz > 0) {
    const dim3 Bq(SCSAMD_BLOCK);
    const int gq = grid_for(nnz);
    hipLaunchKernelGGL(k_tr_count, dim3(gq), Bq, 0, st, ci.p, nnz, rp.p); // rp was zero-filled by alloc
    R.rp.assign((size_t)m + 1, 0);
    rp.download(R.rp.data(), (size_t)m + 1, st);
    HIP_CHECK(hipStreamSynchronize(st));
    eoff maxlen = 0;
    std::vector<int> long_rows;
    for (int i = 0; i < m; ++i) {
      const eoff len = R.rp[i + 1];
      if (len > TR_SHORT) long_rows.push_back(i);
      maxlen = std::max<eoff>(maxlen, len);
      R.rp[i + 1] += R.rp[i];
    }
    if (maxlen <= TR_LONG_MAX) {
      tr_on_dev = true;
      rp.upload(R.rp.data(), (size_t)m + 1, st);
      DevBuf<eoff> nxt((size_t)m + 1);
      DevBuf<int> colidx((size_t)nnz), dlong(long_rows.size() ? long_rows.size() : 1);
      HIP_CHECK(hipMemcpyAsync(nxt.p, rp.p, (size_t)m * sizeof(eoff), hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(k_tr_scatter, dim3(gq), Bq, 0, st, ci.p, nnz, nxt.p, rpos.p);
      hipLaunchKernelGGL(k_tr_sort_short, dim3(grid_for(m)), Bq, 0, st, rp.p, m, rpos.p);
      if (!long_rows.empty()) {
        dlong.upload(long_rows.data(), long_rows.size(), st);
        hipLaunchKernelGGL(k_tr_sort_long, dim3((unsigned)long_rows.size()), Bq, 0, st, rp.p, dlong.p, rpos.p);
      }
      hipLaunchKernelGGL(k_tr_expand_cols, dim3(grid_for(n)), Bq, 0, st, cp.p, n, colidx.p);
      hipLaunchKernelGGL(k_gather_int, dim3(gq), Bq, 0, st, colidx.p, rpos.p, rj.p, nnz);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(st)); // long_rows, nxt, colidx are locals
      R.rj.clear();
      R.pos.clear();
      if (tr_mode == 2) { // verify against the host loop
        CsrPattern H;
        host_transpose_pattern(H);
        std::vector<int> gj((size_t)nnz);
        std::vector<eoff> gp((size_t)nnz);
        rj.download(gj.data(), (size_t)nnz, st);
        rpos.download(gp.data(), (size_t)nnz, st);
        HIP_CHECK(hipStreamSynchronize(st));
        if (H.rp != R.rp || H.rj != gj || H.pos != gp)
          throw HipError("scs_amd: device-built pattern transpose differs from the host's (SCS_AMD_TRANSPOSE=verify)");
      }
    }
  }
  if (!tr_on_dev) {
    host_transpose_pattern(R);
    rp.upload(R.rp.data(), (size_t)m + 1, st);
    rj.upload(R.rj.data(), (size_t)nnz, st);
    rpos.upload(R.pos.data(), (size_t)nnz, st);
  }
  R.built_on_device = tr_on_dev;
  // ---- full symmetric P pattern with the position of each upper entry
  std::vector<eoff> fp, fsrc, upos;
  std::vector<int> fj;
  if (P) {
    fp.assign((size_t)n + 1, 0);
    for (int j = 0; j < n; ++j)
      for (eoff q = P->p[j]; q < P->p[j + 1]; ++q) {
        fp[(size_t)j + 1]++;
        if (P->i[q] != j) fp[(size_t)P->i[q] + 1]++;
      }
    for (int i = 0; i < n; ++i) fp[i + 1] += fp[i];
    std::vector<eoff> nxt(fp.begin(), fp.end() - 1);
    fj.resize((size_t)fp[n]);
    fsrc.resize((size_t)fp[n]);
    upos.resize((size_t)P->p[n]);
    for (int j = 0; j < n; ++j)
      for (eoff q = P->p[j]; q < P->p[j + 1]; ++q) {
        const int i = P->i[q];
        eoff t = nxt[j]++; // row j, column i
        fj[t] = i;
        fsrc[t] = q;
        upos[q] = t;
        if (i != j) {
          t = nxt[i]++; // row i, column j
          fj[t] = j;
          fsrc[t] = q;
        }
      }
  }
  // ---- cone segments after the first (row-wise) block
  const std::vector<int> seg = cone_segments(k);
  std::vector<int> big_off, big_len, small_off, small_len;
  {
    int pos = seg[0];
    for (size_t c = 1; c < seg.size(); ++c) {
      if (seg[c] > SMALL_CONE) {
        big_off.push_back(pos);
        big_len.push_back(seg[c]);
      } else if (seg[c] > 0) {
        small_off.push_back(pos);
        small_len.push_back(seg[c]);
      }
      pos += seg[c];
    }
  }

  // ---- upload
  sc.D.assign((size_t)m, (real)1);
  D.upload(sc.D.data(), (size_t)m, st);
  E.upload(sc.E.data(), (size_t)n, st);
  DevBuf<eoff> pp, psrc, pupos;
  DevBuf<int> pj, boff, blen, soff, slen;
  DevBuf<real> px, pux;
  long long pnnz_full = 0, pnnz_up = 0;
  if (P) {
    pnnz_full = fp[n];
    pnnz_up = P->p[n];
    pp.alloc((size_t)n + 1); pj.alloc((size_t)pnnz_full); psrc.alloc((size_t)pnnz_full);
    pupos.alloc((size_t)pnnz_up); px.alloc((size_t)pnnz_full); pux.alloc((size_t)pnnz_up);
    pp.upload(fp.data(), (size_t)n + 1, st);
    pj.upload(fj.data(), (size_t)pnnz_full, st);
    psrc.upload(fsrc.data(), (size_t)pnnz_full, st);
    pupos.upload(upos.data(), (size_t)pnnz_up, st);
    pux.upload(P->x.data(), (size_t)pnnz_up, st);
    if (pnnz_full)
      hipLaunchKernelGGL(k_gather, dim3(grid_for(pnnz_full)), dim3(SCSAMD_BLOCK), 0, st, pux.p, psrc.p, px.p,
                         pnnz_full);
  }
  if (!big_off.empty()) {
    boff.alloc(big_off.size()); blen.alloc(big_len.size());
    boff.upload(big_off.data(), big_off.size(), st);
    blen.upload(big_len.data(), big_len.size(), st);
  }
  if (!small_off.empty()) {
    soff.alloc(small_off.size()); slen.alloc(small_len.size());
    soff.upload(small_off.data(), small_off.size(), st);
    slen.upload(small_len.data(), small_len.size(), st);
  }
  if (nnz) hipLaunchKernelGGL(k_gather, dim3(grid_for(nnz)), dim3(SCSAMD_BLOCK), 0, st, cx.p, rpos.p, rx.p, nnz);
  HIP_CHECK(hipGetLastError());

  const dim3 B(SCSAMD_BLOCK);
  const int gm = grid_for(m), gn = grid_for(n);
  auto rescale_all = [&] {
    // A <- diag(Dt) A diag(Et) on both orientations, P <- diag(Et) P diag(Et), accumulate
    hipLaunchKernelGGL(k_rescale, dim3(gn), B, 0, st, cp.p, ci.p, cx.p, n, Et.p, Dt.p);
    hipLaunchKernelGGL(k_rescale, dim3(gm), B, 0, st, rp.p, rj.p, rx.p, m, Dt.p, Et.p);
    if (P && pnnz_full) hipLaunchKernelGGL(k_rescale, dim3(gn), B, 0, st, pp.p, pj.p, px.p, n, Et.p, Et.p);
    hipLaunchKernelGGL(k_accumulate, dim3(gm), B, 0, st, D.p, Dt.p, m);
    hipLaunchKernelGGL(k_accumulate, dim3(gn), B, 0, st, E.p, Et.p, n);
  };
  for (int pass = 0; pass < NUM_RUIZ_PASSES; ++pass) { // compute_ruiz_mats :236-307
    hipLaunchKernelGGL(k_row_stat<0>, dim3(gm), B, 0, st, rp.p, rx.p, m, Dt.p);
    if (!big_off.empty())
      hipLaunchKernelGGL(k_cone_max_big, dim3((unsigned)big_off.size()), B, 0, st, boff.p, blen.p, Dt.p);
    if (!small_off.empty())
      hipLaunchKernelGGL(k_cone_max_small, dim3(grid_for((long long)small_off.size())), B, 0, st, soff.p, slen.p,
                         (int)small_off.size(), Dt.p);
    hipLaunchKernelGGL(k_inv_sqrt_scale, dim3(gm), B, 0, st, Dt.p, m);
    hipLaunchKernelGGL(k_col_stat<0>, dim3(gn), B, 0, st, cp.p, cx.p, P ? (const eoff *)pp.p : (const eoff *)nullptr,
                       P ? px.p : (const real *)nullptr, n, Et.p);
    rescale_all();
  }
  HIP_CHECK(hipGetLastError());
  std::vector<real> hDt((size_t)m);
  for (int pass = 0; pass < NUM_L2_PASSES; ++pass) { // compute_l2_mats :309-368
    hipLaunchKernelGGL(k_row_stat<1>, dim3(gm), B, 0, st, rp.p, rx.p, m, Dt.p);
    // ordered per-cone mean on the host (see header), then 1/sqrt(clip(.)) there too
    Dt.download(hDt.data(), (size_t)m, st);
    HIP_CHECK(hipStreamSynchronize(st));
    {
      size_t pos = (size_t)seg[0];
      for (size_t c = 1; c < seg.size(); ++c) {
        const int len = seg[c];
        real w = 0;
        if (len > 0) {
          for (int j = 0; j < len; ++j) w += hDt[pos + j];
          w /= (real)len;
        }
        for (int j = 0; j < len; ++j) hDt[pos + j] = w;
        pos += (size_t)len;
      }
    }
    Dt.upload(hDt.data(), (size_t)m, st);
    hipLaunchKernelGGL(k_inv_sqrt_scale, dim3(gm), B, 0, st, Dt.p, m);
    hipLaunchKernelGGL(k_col_stat<1>, dim3(gn), B, 0, st, cp.p, cx.p, P ? (const eoff *)pp.p : (const eoff *)nullptr,
                       P ? px.p : (const real *)nullptr, n, Et.p);
    rescale_all();
    HIP_CHECK(hipStreamSynchronize(st)); // hDt is reused / freed
  }
  // ---- results back to the host copies
  cx.download(A.x.data(), (size_t)nnz, st);
  D.download(sc.D.data(), (size_t)m, st);
  E.download(sc.E.data(), (size_t)n, st);
  if (P && pnnz_up) {
    hipLaunchKernelGGL(k_gather, dim3(grid_for(pnnz_up)), dim3(SCSAMD_BLOCK), 0, st, px.p, pupos.p, pux.p, pnnz_up);
    pux.download(P->x.data(), (size_t)pnnz_up, st);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));
  if (csr_cache) { // both orientations stay in HBM for LinSys::init (values: bit-identical to gathering the CSC values through the map)
    DevMatrices &K = csr_cache->dev;
    K.cp.take(cp);
    K.ci.take(ci);
    K.cx.take(cx);
    K.rp.take(rp);
    K.rj.take(rj);
    K.rx.take(rx);
    K.rpos.take(rpos);
    K.valid = true;
  }
}

} // namespace scsamd
