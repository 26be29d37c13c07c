// lab/spmv_lab.hip -- micro-experiments behind the SpMV design (not product code).
// Builds a cfg2-shaped random matrix (n cols, m=2n rows, col_nnz per column) and
// times: (1) a pure streaming copy, (2) random 8-byte gathers from tables of
// varying footprint, (3) the csr_stream kernel variants in both orientations.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 lab/spmv_lab.hip -o lab/spmv_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <typename T> T *dev(const std::vector<T> &h) {
  T *p; CK(hipMalloc(&p, (h.size() + 64) * sizeof(T)));
  CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}
template <typename T> T *devz(size_t n) { T *p; CK(hipMalloc(&p, (n + 64) * sizeof(T))); CK(hipMemset(p, 0, (n + 64) * sizeof(T))); return p; }

template <typename F> double time_us(F f, int reps = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return 1e3 * ms / reps;
}

// ---- (1) streaming copy ------------------------------------------------------
__global__ void k_copy(const double2 *__restrict__ a, double2 *b, size_t n2) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// ---- (1b) stream val+idx and reduce (no gather) ----------------------------------
template <int NT> __global__ void k_stream(const double *__restrict__ val, const int *__restrict__ idx, double *out, size_t nnz) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x) {
    double v = NT ? __builtin_nontemporal_load(val + i) : val[i];
    int c = NT ? __builtin_nontemporal_load(idx + i) : idx[i];
    acc += v * (double)(c & 1);
  }
  if (acc == 12345.678) out[0] = acc;
}
// ---- (2) random gathers from a table of `tab` doubles ---------------------------
template <int NT, int U> __global__ void k_gather(const int *__restrict__ idx, const double *__restrict__ x, double *out, size_t nnz, int mask) {
  double acc = 0;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + (U - 1) * stride < nnz; i += U * stride) {
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = (NT ? __builtin_nontemporal_load(idx + i + u * stride) : idx[i + u * stride]) & mask;
#pragma unroll
    for (int u = 0; u < U; ++u) acc += x[c[u]];
  }
  if (acc == 12345.678) out[0] = acc;
}

// cache-policy variants of the 8-byte gather: 0 plain, 1 nt builtin, 2 sc1, 3 sc0 sc1, 4 nt asm
template <int POL> __device__ __forceinline__ double gload(const double *p) {
  if (POL == 0) return *p;
  if (POL == 1) return __builtin_nontemporal_load(p);
  double v;
  if (POL == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POL == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (POL == 4) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int POL> __global__ void k_gather_pol(const int *__restrict__ idx, const double *x, double *out, size_t nnz, int mask) {
  double acc = 0;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += stride) acc += gload<POL>(x + (idx[i] & mask));
  if (acc == 12345.678) out[0] = acc;
}

// ---- (3) csr_stream variants --------------------------------------------------------
struct Csr { int rows, cols, nblk; const int *ptr, *idx, *rowblk; const double *val; };
// FLAGS: bit0 = nontemporal streams, bit1 = skip gather (x[0]), bit2 = skip phase 2
template <int NNZB, int BLOCK, int FLAGS>
__global__ __launch_bounds__(BLOCK) void k_csr_stream(Csr A, const double *__restrict__ x, double *y) {
  __shared__ double prod[NNZB];
  constexpr int U = NNZB / BLOCK;
  const int tid = threadIdx.x;
  for (int b = blockIdx.x; b < A.nblk; b += gridDim.x) {
    const int r0 = A.rowblk[b], r1 = A.rowblk[b + 1];
    const int k0 = A.ptr[r0], cnt = A.ptr[r1] - k0;
    int ii[U]; double vv[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int k = tid + j * BLOCK; const bool ok = k < cnt;
      if (FLAGS & 1) { ii[j] = ok ? __builtin_nontemporal_load(A.idx + k0 + k) : 0; vv[j] = ok ? __builtin_nontemporal_load(A.val + k0 + k) : 0.0; }
      else { ii[j] = ok ? A.idx[k0 + k] : 0; vv[j] = ok ? A.val[k0 + k] : 0.0; }
    }
    double xx[U];
#pragma unroll
    for (int j = 0; j < U; ++j) xx[j] = (FLAGS & 2) ? x[ii[j] & 1] : x[ii[j]];
#pragma unroll
    for (int j = 0; j < U; ++j) { const int k = tid + j * BLOCK; if (k < cnt) prod[k] = vv[j] * xx[j]; }
    __syncthreads();
    if (!(FLAGS & 4)) {
      for (int r = r0 + tid; r < r1; r += BLOCK) {
        const int a = A.ptr[r] - k0, z = A.ptr[r + 1] - k0;
        double acc = 0;
        for (int k = a; k < z; ++k) acc += prod[k];
        if (FLAGS & 8) y[r] += acc; else y[r] = acc;
      }
    } else if (tid == 0) y[r0] = prod[0];
    __syncthreads();
  }
}

// wave-per-chunk variant without LDS: each lane owns a contiguous run of rows? (row-per-lane, "scalar CSR")
__global__ __launch_bounds__(256) void k_csr_scalar(Csr A, const double *__restrict__ x, double *y) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < A.rows; r += gridDim.x * blockDim.x) {
    double acc = 0;
    const int a = A.ptr[r], z = A.ptr[r + 1];
    for (int k = a; k < z; ++k) acc += A.val[k] * x[A.idx[k]];
    y[r] = acc;
  }
}


// ---- (4) column-sliced, time-aligned SpMV ("all workgroups walk the slices of x together") ----
struct Sliced { int rows, nsb, S, SB; const int *sbrow; const int *segoff; const unsigned *sidx; const double *sval; };
template <int CH, int NT = 0> __global__ __launch_bounds__(256) void k_sliced(Sliced A, const double *__restrict__ x, double *y, int accrows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *acc = reinterpret_cast<double *>(smem);
  double *sp = acc + accrows;
  unsigned short *sr = reinterpret_cast<unsigned short *>(sp + CH);
  const int tid = threadIdx.x, b = blockIdx.x;
  const int r0 = A.sbrow[b], nr = A.sbrow[b + 1] - r0;
  for (int r = tid; r < nr; r += 256) acc[r] = 0;
  __syncthreads();
  const unsigned mask = (1u << A.SB) - 1;
  for (int s = 0; s < A.S; ++s) {
    const int e0 = A.segoff[b * (A.S + 1) + s], e1 = A.segoff[b * (A.S + 1) + s + 1];
    const double *xs = x + ((size_t)s << A.SB);
    for (int base = e0; base < e1; base += CH) {
      const int cnt = min(CH, e1 - base);
      unsigned w[CH / 256]; double v[CH / 256];
#pragma unroll
      for (int j = 0; j < CH / 256; ++j) { const int k = tid + j * 256; const bool ok = k < cnt;
        if (NT) { w[j] = ok ? __builtin_nontemporal_load(A.sidx + base + k) : 0u; v[j] = ok ? __builtin_nontemporal_load(A.sval + base + k) : 0.0; }
        else { w[j] = ok ? A.sidx[base + k] : 0u; v[j] = ok ? A.sval[base + k] : 0.0; } }
      double xx[CH / 256];
#pragma unroll
      for (int j = 0; j < CH / 256; ++j) xx[j] = xs[w[j] & mask];
#pragma unroll
      for (int j = 0; j < CH / 256; ++j) { const int k = tid + j * 256; if (k < cnt) { sp[k] = v[j] * xx[j]; sr[k] = (unsigned short)(w[j] >> A.SB); } }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < CH / 256; ++j) { const int k = tid + j * 256;
        if (k < cnt) { const unsigned short lr = sr[k];
          if (k == 0 || sr[k - 1] != lr) { double sum = sp[k]; int kk = k + 1; while (kk < cnt && sr[kk] == lr) sum += sp[kk++]; acc[lr] += sum; } } }
      __syncthreads();
    }
  }
  for (int r = tid; r < nr; r += 256) y[r0 + r] = acc[r];
}


// pipelined variant: the next chunk's idx/val loads are in flight while the current chunk is reduced
template <int CH> __global__ __launch_bounds__(256) void k_sliced_pf(Sliced A, const double *__restrict__ x, double *y, int accrows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *acc = reinterpret_cast<double *>(smem);
  double *sp = acc + accrows;
  unsigned short *sr = reinterpret_cast<unsigned short *>(sp + CH);
  constexpr int U = CH / 256;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int r0 = A.sbrow[b], nr = A.sbrow[b + 1] - r0;
  for (int r = tid; r < nr; r += 256) acc[r] = 0;
  const unsigned mask = (1u << A.SB) - 1;
  const int *so = A.segoff + (size_t)b * (A.S + 1);
  const int eend = so[A.S];
  // chunk cursor: [base, base+cnt) inside slice s
  int s = 0, base = so[0];
  while (s < A.S && base >= so[s + 1]) ++s;
  unsigned w[U]; double v[U];
  int cnt = 0;
  if (s < A.S) { cnt = min(CH, so[s + 1] - base);
#pragma unroll
    for (int j = 0; j < U; ++j) { const int k = tid + j * 256; const bool ok = k < cnt; w[j] = ok ? A.sidx[base + k] : 0u; v[j] = ok ? A.sval[base + k] : 0.0; } }
  __syncthreads();
  while (s < A.S) {
    const double *xs = x + ((size_t)s << A.SB);
    double xx[U];
#pragma unroll
    for (int j = 0; j < U; ++j) xx[j] = xs[w[j] & mask];
    // cursor of the next chunk + its loads (prefetch)
    int s2 = s, base2 = base + cnt;
    while (s2 < A.S && base2 >= so[s2 + 1]) ++s2;
    unsigned w2[U]; double v2[U]; int cnt2 = 0;
    if (s2 < A.S) { cnt2 = min(CH, so[s2 + 1] - base2);
#pragma unroll
      for (int j = 0; j < U; ++j) { const int k = tid + j * 256; const bool ok = k < cnt2; w2[j] = ok ? A.sidx[base2 + k] : 0u; v2[j] = ok ? A.sval[base2 + k] : 0.0; } }
#pragma unroll
    for (int j = 0; j < U; ++j) { const int k = tid + j * 256; if (k < cnt) { sp[k] = v[j] * xx[j]; sr[k] = (unsigned short)(w[j] >> A.SB); } }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < U; ++j) { const int k = tid + j * 256;
      if (k < cnt) { const unsigned short lr = sr[k];
        if (k == 0 || sr[k - 1] != lr) { double sum = sp[k]; int kk = k + 1; while (kk < cnt && sr[kk] == lr) sum += sp[kk++]; acc[lr] += sum; } } }
    __syncthreads();
    s = s2; base = base2; cnt = cnt2;
#pragma unroll
    for (int j = 0; j < U; ++j) { w[j] = w2[j]; v[j] = v2[j]; }
  }
  (void)eend;
  for (int r = tid; r < nr; r += 256) y[r0 + r] = acc[r];
}


// wave-per-super-block variant: no workgroup barriers at all (a wave's LDS ops are in order)
template <int CHW, int ACCW> __global__ __launch_bounds__(256) void k_sliced_wave(Sliced A, const double *__restrict__ x, double *y) {
  __shared__ double acc_all[4][ACCW];
  __shared__ double sp_all[4][CHW];
  __shared__ unsigned short sr_all[4][CHW];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wv;
  if (b >= A.nsb) return;
  double *acc = acc_all[wv]; double *sp = sp_all[wv]; unsigned short *sr = sr_all[wv];
  constexpr int U = CHW / 64;
  const int r0 = A.sbrow[b], nr = A.sbrow[b + 1] - r0;
  for (int r = lane; r < nr; r += 64) acc[r] = 0;
  const unsigned mask = (1u << A.SB) - 1;
  const int *so = A.segoff + (size_t)b * (A.S + 1);
  for (int s = 0; s < A.S; ++s) {
    const int e0 = so[s], e1 = so[s + 1];
    const double *xs = x + ((size_t)s << A.SB);
    for (int base = e0; base < e1; base += CHW) {
      const int cnt = min(CHW, e1 - base);
      unsigned w[U]; double v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) { const int k = lane + j * 64; const bool ok = k < cnt; w[j] = ok ? A.sidx[base + k] : 0u; v[j] = ok ? A.sval[base + k] : 0.0; }
      double xx[U];
#pragma unroll
      for (int j = 0; j < U; ++j) xx[j] = xs[w[j] & mask];
#pragma unroll
      for (int j = 0; j < U; ++j) { const int k = lane + j * 64; if (k < cnt) { sp[k] = v[j] * xx[j]; sr[k] = (unsigned short)(w[j] >> A.SB); } }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): LDS writes of this wave are visible to its lanes
#pragma unroll
      for (int j = 0; j < U; ++j) { const int k = lane + j * 64;
        if (k < cnt) { const unsigned short lr = sr[k];
          if (k == 0 || sr[k - 1] != lr) { double sum = sp[k]; int kk = k + 1; while (kk < cnt && sr[kk] == lr) sum += sp[kk++]; acc[lr] += sum; } } }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
  }
  for (int r = lane; r < nr; r += 64) y[r0 + r] = acc[r];
}


// two-deep software pipeline: gathers of chunk c+1 and the idx/val loads of chunk c+2 are in
// flight while chunk c is reduced in LDS
template <int CH> __global__ __launch_bounds__(256) void k_sliced_pf2(Sliced A, const double *__restrict__ x, double *y, int accrows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *acc = reinterpret_cast<double *>(smem);
  double *sp = acc + accrows;
  unsigned short *sr = reinterpret_cast<unsigned short *>(sp + CH);
  constexpr int U = CH / 256;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int r0 = A.sbrow[b], nr = A.sbrow[b + 1] - r0;
  for (int r = tid; r < nr; r += 256) acc[r] = 0;
  const unsigned mask = (1u << A.SB) - 1;
  const int *so = A.segoff + (size_t)b * (A.S + 1);
  // cursor helper: advance (s, base) past the chunk [base, base+cnt)
  auto advance = [&](int &s, int &base, int cnt) { base += cnt; while (s < A.S && base >= so[s + 1]) ++s; };
  auto chunk_cnt = [&](int s, int base) { return s < A.S ? min(CH, so[s + 1] - base) : 0; };
  int s0 = 0, b0 = so[0]; while (s0 < A.S && b0 >= so[s0 + 1]) ++s0;
  int c0 = chunk_cnt(s0, b0);
  unsigned w0[U]; double v0[U], x0[U];
#pragma unroll
  for (int j = 0; j < U; ++j) { const int k = tid + j * 256; const bool ok = k < c0; w0[j] = ok ? A.sidx[b0 + k] : 0u; v0[j] = ok ? A.sval[b0 + k] : 0.0; }
  int s1 = s0, b1 = b0; advance(s1, b1, c0);
  int c1 = chunk_cnt(s1, b1);
  unsigned w1[U]; double v1[U];
#pragma unroll
  for (int j = 0; j < U; ++j) { const int k = tid + j * 256; const bool ok = k < c1; w1[j] = ok ? A.sidx[b1 + k] : 0u; v1[j] = ok ? A.sval[b1 + k] : 0.0; }
  {
    const double *xs = x + ((size_t)(s0 < A.S ? s0 : 0) << A.SB);
#pragma unroll
    for (int j = 0; j < U; ++j) x0[j] = xs[w0[j] & mask];
  }
  __syncthreads();
  while (c0 > 0) {
    // issue gathers for chunk 1 and loads for chunk 2 BEFORE consuming chunk 0
    double x1[U];
    {
      const double *xs1 = x + ((size_t)(s1 < A.S ? s1 : 0) << A.SB);
#pragma unroll
      for (int j = 0; j < U; ++j) x1[j] = xs1[w1[j] & mask];
    }
    int s2 = s1, b2 = b1; advance(s2, b2, c1);
    const int c2 = chunk_cnt(s2, b2);
    unsigned w2[U]; double v2[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { const int k = tid + j * 256; const bool ok = k < c2; w2[j] = ok ? A.sidx[b2 + k] : 0u; v2[j] = ok ? A.sval[b2 + k] : 0.0; }
#pragma unroll
    for (int j = 0; j < U; ++j) { const int k = tid + j * 256; if (k < c0) { sp[k] = v0[j] * x0[j]; sr[k] = (unsigned short)(w0[j] >> A.SB); } }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < U; ++j) { const int k = tid + j * 256;
      if (k < c0) { const unsigned short lr = sr[k];
        if (k == 0 || sr[k - 1] != lr) { double sum = sp[k]; int kk = k + 1; while (kk < c0 && sr[kk] == lr) sum += sp[kk++]; acc[lr] += sum; } } }
    __syncthreads();
    c0 = c1; s0 = s1; b0 = b1;
    c1 = c2; s1 = s2; b1 = b2;
#pragma unroll
    for (int j = 0; j < U; ++j) { w0[j] = w1[j]; v0[j] = v1[j]; x0[j] = x1[j]; w1[j] = w2[j]; v1[j] = v2[j]; }
  }
  for (int r = tid; r < nr; r += 256) y[r0 + r] = acc[r];
}

static int g_maxrows = 0;
static Sliced build_sliced(int rows, int cols, const std::vector<int> &ptr, const std::vector<int> &idx, const std::vector<double> &val, int SB, int nnz_sb) {
  int S = (cols + (1 << SB) - 1) >> SB;
  std::vector<int> sbrow; sbrow.push_back(0);
  int r = 0;
  while (r < rows) { int s0 = r; long long acc = 0; while (r < rows && r - s0 < 4096) { long long rn = ptr[r + 1] - ptr[r]; if (acc + rn > nnz_sb && r > s0) break; acc += rn; ++r; } sbrow.push_back(r); }
  int nsb = (int)sbrow.size() - 1;
  g_maxrows = 0; for (int b = 0; b < nsb; ++b) g_maxrows = std::max(g_maxrows, sbrow[b + 1] - sbrow[b]);
  size_t nnz = idx.size();
  std::vector<unsigned> sidx(nnz); std::vector<double> sval(nnz); std::vector<int> segoff((size_t)nsb * (S + 1));
  for (int b = 0; b < nsb; ++b) {
    int k0 = ptr[sbrow[b]], k1 = ptr[sbrow[b + 1]];
    std::vector<int> cnt(S + 1, 0);
    for (int k = k0; k < k1; ++k) cnt[(idx[k] >> SB) + 1]++;
    for (int s = 0; s < S; ++s) cnt[s + 1] += cnt[s];
    for (int s = 0; s <= S; ++s) segoff[(size_t)b * (S + 1) + s] = k0 + cnt[s];
    std::vector<int> nx(cnt.begin(), cnt.end() - 1);
    for (int rr = sbrow[b]; rr < sbrow[b + 1]; ++rr) for (int k = ptr[rr]; k < ptr[rr + 1]; ++k) {
      int s = idx[k] >> SB; int q = k0 + nx[s]++; sidx[q] = (unsigned)(idx[k] & ((1 << SB) - 1)) | ((unsigned)(rr - sbrow[b]) << SB); sval[q] = val[k]; }
  }
  Sliced A{rows, nsb, S, SB, dev(sbrow), dev(segoff), dev(sidx), dev(sval)};
  return A;
}

static void build_rowblk(int rows, const std::vector<int> &ptr, int nnzb, int rowsmax, std::vector<int> &rb) {
  rb.clear(); rb.push_back(0); int r = 0;
  while (r < rows) { int s = r; long long acc = 0;
    while (r < rows && r - s < rowsmax) { long long rn = ptr[r + 1] - ptr[r]; if (acc + rn > nnzb) break; acc += rn; ++r; }
    if (r == s) ++r; rb.push_back(r); }
}

int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 1000000; int cn = argc > 2 ? atoi(argv[2]) : 10; int m = 2 * n;
  size_t nnz = (size_t)n * cn;
  printf("n=%d m=%d nnz=%zu\n", n, m, nnz);
  std::mt19937_64 rng(1);
  // CSC(A) = CSR(A'): n rows, exactly cn sorted distinct entries each, cols in [0,m)
  std::vector<int> Tp(n + 1), Ti(nnz); std::vector<double> Tx(nnz);
  for (int j = 0; j < n; ++j) { Tp[j] = j * cn; int *r = &Ti[(size_t)j * cn];
    for (;;) { for (int k = 0; k < cn; ++k) r[k] = (int)(rng() % m); std::sort(r, r + cn); if (std::adjacent_find(r, r + cn) == r + cn) break; }
    for (int k = 0; k < cn; ++k) Tx[(size_t)j * cn + k] = (double)(rng() % 2001) / 1000.0 - 1.0; }
  Tp[n] = (int)nnz;
  // transpose -> CSR(A): m rows
  std::vector<int> Ap(m + 1, 0), Ai(nnz); std::vector<double> Ax(nnz);
  for (size_t k = 0; k < nnz; ++k) Ap[Ti[k] + 1]++;
  for (int i = 0; i < m; ++i) Ap[i + 1] += Ap[i];
  { std::vector<int> nx(Ap.begin(), Ap.end() - 1);
    for (int j = 0; j < n; ++j) for (int k = Tp[j]; k < Tp[j + 1]; ++k) { int q = nx[Ti[k]]++; Ai[q] = j; Ax[q] = Tx[k]; } }
  std::vector<double> hx(m); for (auto &v : hx) v = (double)(rng() % 1000) / 1000.0;

  int *dTp = dev(Tp), *dTi = dev(Ti), *dAp = dev(Ap), *dAi = dev(Ai);
  double *dTx = dev(Tx), *dAx = dev(Ax), *dxm = dev(hx), *dy = devz<double>(m), *dout = devz<double>(16);

  // (1) copy
  double *dcopy = devz<double>(nnz);
  { size_t n2 = nnz / 2; double us = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, (const double2 *)dTx, (double2 *)dcopy, n2); });
    printf("copy 2x%.0f MB: %.1f us  %.0f GB/s\n", nnz * 8 / 1e6, us, 2.0 * nnz * 8 / us / 1e3); }
  { double us = time_us([&] { hipLaunchKernelGGL((k_stream<0>), dim3(4096), dim3(256), 0, 0, dTx, dTi, dout, nnz); });
    printf("stream val+idx (120 MB): %.1f us  %.0f GB/s\n", us, nnz * 12.0 / us / 1e3);
    us = time_us([&] { hipLaunchKernelGGL((k_stream<1>), dim3(4096), dim3(256), 0, 0, dTx, dTi, dout, nnz); });
    printf("stream val+idx nt      : %.1f us  %.0f GB/s\n", us, nnz * 12.0 / us / 1e3); }
  // (2) gathers vs footprint
  for (int lg = 17; lg <= 21; ++lg) { int mask = (1 << lg) - 1; if (mask >= m) mask = (1 << 20) - 1;
    double us = time_us([&] { hipLaunchKernelGGL((k_gather<0, 8>), dim3(4096), dim3(256), 0, 0, dTi, dxm, dout, nnz, mask); });
    double us2 = time_us([&] { hipLaunchKernelGGL((k_gather<1, 8>), dim3(4096), dim3(256), 0, 0, dTi, dxm, dout, nnz, mask); });
    printf("gather 1e7 x 8B from %5.1f MB table: %.1f us (%.1f Ggather/s) ; nt idx %.1f us\n", (mask + 1) * 8 / 1e6, us, nnz / us / 1e3, us2); }
  { double us = time_us([&] { hipLaunchKernelGGL((k_gather<0, 8>), dim3(4096), dim3(256), 0, 0, dTi, dxm, dout, nnz, 0x7fffffff); });
    printf("gather 1e7 x 8B from 16 MB (full m) table: %.1f us (%.1f Ggather/s)\n", us, nnz / us / 1e3); }

  for (int mask : {(1 << 17) - 1, (1 << 20) - 1, 0x7fffffff}) {
    double t[5];
    t[0] = time_us([&] { hipLaunchKernelGGL((k_gather_pol<0>), dim3(8192), dim3(256), 0, 0, dTi, dxm, dout, nnz, mask); });
    t[1] = time_us([&] { hipLaunchKernelGGL((k_gather_pol<1>), dim3(8192), dim3(256), 0, 0, dTi, dxm, dout, nnz, mask); });
    t[2] = time_us([&] { hipLaunchKernelGGL((k_gather_pol<2>), dim3(8192), dim3(256), 0, 0, dTi, dxm, dout, nnz, mask); });
    t[3] = time_us([&] { hipLaunchKernelGGL((k_gather_pol<3>), dim3(8192), dim3(256), 0, 0, dTi, dxm, dout, nnz, mask); });
    t[4] = time_us([&] { hipLaunchKernelGGL((k_gather_pol<4>), dim3(8192), dim3(256), 0, 0, dTi, dxm, dout, nnz, mask); });
    printf("gather policy (mask %x) [1 load in flight/lane]: plain %.1f  nt %.1f  sc1 %.1f  sc0sc1 %.1f  nt-asm %.1f us\n", mask, t[0], t[1], t[2], t[3], t[4]);
  }
  // (2c) column-sliced passes: S sub-matrices of CSR(A) by column range, y += A_s x_s per pass
  for (int S : {2, 4, 8}) {
    std::vector<std::vector<int>> sp(S, std::vector<int>(m + 1, 0)), si(S);
    std::vector<std::vector<double>> sx(S);
    for (int r = 0; r < m; ++r) {
      for (int k = Ap[r]; k < Ap[r + 1]; ++k) { int sl = (int)((long long)Ai[k] * S / n); si[sl].push_back(Ai[k]); sx[sl].push_back(Ax[k]); }
      for (int t = 0; t < S; ++t) sp[t][r + 1] = (int)si[t].size();
    }
    std::vector<Csr> mats; std::vector<int> grids;
    for (int t = 0; t < S; ++t) { std::vector<int> rb; build_rowblk(m, sp[t], 2048, 2048, rb);
      Csr A{m, n, (int)rb.size() - 1, dev(sp[t]), dev(si[t]), dev(rb), dev(sx[t])}; mats.push_back(A); grids.push_back(std::min(A.nblk, 16384)); }
    double us = time_us([&] { for (int t = 0; t < S; ++t) hipLaunchKernelGGL((k_csr_stream<2048, 256, 8>), dim3(grids[t]), dim3(256), 0, 0, mats[t], dxm, dy); });
    printf("A column-sliced S=%d passes (y += A_s x_s): %.1f us total\n", S, us);
  }

  // (4) sliced kernels, correctness vs scalar kernel + timing
  {
    double *dref = devz<double>(m), *dys = devz<double>(m);
    auto check = [&](const char *name, int rows, int cols, const std::vector<int> &P, const std::vector<int> &I, const std::vector<double> &V, int *dp, int *di, double *dv, long long bytes) {
      Csr C{rows, cols, 0, dp, di, nullptr, dv};
      hipLaunchKernelGGL(k_csr_scalar, dim3(8192), dim3(256), 0, 0, C, dxm, dref);
      for (int SB : {16, 17, 18}) for (int nnz_sb : {8192}) {
        Sliced A = build_sliced(rows, cols, P, I, V, SB, nnz_sb);
        int ar = (g_maxrows + 1) & ~1;
        size_t l2 = (size_t)ar * 8 + 2048 * 10, l1 = (size_t)ar * 8 + 1024 * 10, l0 = (size_t)ar * 8 + 512 * 10;
        CK(hipMemset(dys, 0, rows * 8));
        hipLaunchKernelGGL((k_sliced<2048>), dim3(A.nsb), dim3(256), l2, 0, A, dxm, dys, ar);
        std::vector<double> h1(rows), h2(rows); CK(hipMemcpy(h1.data(), dref, rows * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), dys, rows * 8, hipMemcpyDeviceToHost));
        double err = 0; for (int i = 0; i < rows; ++i) err = std::max(err, fabs(h1[i] - h2[i]));
        double us = time_us([&] { hipLaunchKernelGGL((k_sliced<2048>), dim3(A.nsb), dim3(256), l2, 0, A, dxm, dys, ar); });
        double us1 = time_us([&] { hipLaunchKernelGGL((k_sliced<1024>), dim3(A.nsb), dim3(256), l1, 0, A, dxm, dys, ar); });
        double us0 = time_us([&] { hipLaunchKernelGGL((k_sliced<512>), dim3(A.nsb), dim3(256), l0, 0, A, dxm, dys, ar); });
        double p2 = time_us([&] { hipLaunchKernelGGL((k_sliced_pf<2048>), dim3(A.nsb), dim3(256), l2, 0, A, dxm, dys, ar); });
        double p1 = time_us([&] { hipLaunchKernelGGL((k_sliced_pf<1024>), dim3(A.nsb), dim3(256), l1, 0, A, dxm, dys, ar); });
        double p0 = time_us([&] { hipLaunchKernelGGL((k_sliced_pf<512>), dim3(A.nsb), dim3(256), l0, 0, A, dxm, dys, ar); });
        CK(hipMemset(dys, 0, rows * 8));
        hipLaunchKernelGGL((k_sliced_pf<1024>), dim3(A.nsb), dim3(256), l1, 0, A, dxm, dys, ar);
        { std::vector<double> h3(rows); CK(hipMemcpy(h3.data(), dys, rows * 8, hipMemcpyDeviceToHost)); double e3 = 0; for (int i = 0; i < rows; ++i) e3 = std::max(e3, fabs(h1[i] - h3[i])); printf("   PF: CH2048 %.1f | CH1024 %.1f | CH512 %.1f us  err %.1e\n", p2, p1, p0, e3); }
        { CK(hipMemset(dys, 0, rows * 8));
          hipLaunchKernelGGL((k_sliced_pf2<512>), dim3(A.nsb), dim3(256), l0, 0, A, dxm, dys, ar);
          std::vector<double> h3(rows); CK(hipMemcpy(h3.data(), dys, rows * 8, hipMemcpyDeviceToHost)); double e3 = 0; for (int i = 0; i < rows; ++i) e3 = std::max(e3, fabs(h1[i] - h3[i]));
          double q0 = time_us([&] { hipLaunchKernelGGL((k_sliced_pf2<256>), dim3(A.nsb), dim3(256), l0, 0, A, dxm, dys, ar); });
          double q1 = time_us([&] { hipLaunchKernelGGL((k_sliced_pf2<512>), dim3(A.nsb), dim3(256), l0, 0, A, dxm, dys, ar); });
          double q2 = time_us([&] { hipLaunchKernelGGL((k_sliced_pf2<1024>), dim3(A.nsb), dim3(256), l1, 0, A, dxm, dys, ar); });
          printf("   PF2 (2-deep pipeline): CH256 %.1f | CH512 %.1f | CH1024 %.1f us  err %.1e\n", q0, q1, q2, e3); }
        { double n1 = time_us([&] { hipLaunchKernelGGL((k_sliced<1024, 1>), dim3(A.nsb), dim3(256), l1, 0, A, dxm, dys, ar); });
          double n0 = time_us([&] { hipLaunchKernelGGL((k_sliced<512, 1>), dim3(A.nsb), dim3(256), l0, 0, A, dxm, dys, ar); });
          printf("   NT streams: CH1024 %.1f | CH512 %.1f us\n", n1, n0); }
        printf("%s sliced SB=%d (S=%d) nnz_sb=%d nsb=%d maxrows=%d: CH2048 %.1f us | CH1024 %.1f us | CH512 %.1f us (%.0f GB/s best) maxerr %.2e\n", name, SB, A.S, nnz_sb, A.nsb, g_maxrows, us, us1, us0, bytes / std::min(us, std::min(us1, us0)) / 1e3, err);
        CK(hipFree((void*)A.sbrow)); CK(hipFree((void*)A.segoff)); CK(hipFree((void*)A.sidx)); CK(hipFree((void*)A.sval));
      }
    };
    long long bA2 = (long long)nnz * 12 + (m + 1) * 4LL + n * 8LL + m * 8LL, bT2 = (long long)nnz * 12 + (n + 1) * 4LL + m * 8LL + n * 8LL;

    auto checkw = [&](const char *name, int rows, int cols, const std::vector<int> &P, const std::vector<int> &I, const std::vector<double> &V, int *dp, int *di, double *dv, long long bytes) {
      Csr C{rows, cols, 0, dp, di, nullptr, dv};
      hipLaunchKernelGGL(k_csr_scalar, dim3(8192), dim3(256), 0, 0, C, dxm, dref);
      std::vector<double> h1(rows), h2(rows); CK(hipMemcpy(h1.data(), dref, rows * 8, hipMemcpyDeviceToHost));
      for (int SB : {16, 17}) for (int nnz_sb : {1024, 2048, 4096}) {
        Sliced A = build_sliced(rows, cols, P, I, V, SB, nnz_sb);
        if (g_maxrows > 1024) { printf("%s wave SB=%d nnz_sb=%d: maxrows %d too large\n", name, SB, nnz_sb, g_maxrows); continue; }
        int g = (A.nsb + 3) / 4;
        CK(hipMemset(dys, 0, rows * 8));
        hipLaunchKernelGGL((k_sliced_wave<128, 1024>), dim3(g), dim3(256), 0, 0, A, dxm, dys);
        CK(hipMemcpy(h2.data(), dys, rows * 8, hipMemcpyDeviceToHost));
        double err = 0; for (int i = 0; i < rows; ++i) err = std::max(err, fabs(h1[i] - h2[i]));
        double u1 = time_us([&] { hipLaunchKernelGGL((k_sliced_wave<128, 1024>), dim3(g), dim3(256), 0, 0, A, dxm, dys); });
        double u2 = time_us([&] { hipLaunchKernelGGL((k_sliced_wave<256, 1024>), dim3(g), dim3(256), 0, 0, A, dxm, dys); });
        double u3 = time_us([&] { hipLaunchKernelGGL((k_sliced_wave<64, 1024>), dim3(g), dim3(256), 0, 0, A, dxm, dys); });
        printf("%s WAVE SB=%d (S=%d) nnz_sb=%d nsb=%d maxrows=%d: CHW128 %.1f | CHW256 %.1f | CHW64 %.1f us (%.0f GB/s best) err %.1e\n", name, SB, A.S, nnz_sb, A.nsb, g_maxrows, u1, u2, u3, bytes / std::min(u1, std::min(u2, u3)) / 1e3, err);
        CK(hipFree((void*)A.sbrow)); CK(hipFree((void*)A.segoff)); CK(hipFree((void*)A.sidx)); CK(hipFree((void*)A.sval));
      }
    };
    check("A ", m, n, Ap, Ai, Ax, dAp, dAi, dAx, bA2);
    check("At", n, m, Tp, Ti, Tx, dTp, dTi, dTx, bT2);
    if (getenv("LAB_WAVE")) checkw("A ", m, n, Ap, Ai, Ax, dAp, dAi, dAx, bA2);
    if (getenv("LAB_WAVE")) checkw("At", n, m, Tp, Ti, Tx, dTp, dTi, dTx, bT2);

  }
  // (3) csr_stream variants
  auto run = [&](const char *name, int rows, int cols, int *p, int *i, double *v, double *x, long long bytes) {
    std::vector<int> hp(rows + 1); CK(hipMemcpy(hp.data(), p, (rows + 1) * 4, hipMemcpyDeviceToHost));
    for (int nnzb : {1024, 2048, 4096}) {
      std::vector<int> rb; build_rowblk(rows, hp, nnzb, 2048, rb); int *drb = dev(rb);
      Csr A{rows, cols, (int)rb.size() - 1, p, i, drb, v};
      int g = std::min(A.nblk, 16384);
#define RUN(NB, BL, FL, label) if (nnzb == NB) { double us = time_us([&] { hipLaunchKernelGGL((k_csr_stream<NB, BL, FL>), dim3(g), dim3(BL), 0, 0, A, x, dy); }); \
        printf("%s nnzb=%d block=%d %-22s: %7.1f us  %6.0f GB/s\n", name, NB, BL, label, us, bytes / us / 1e3); }
      RUN(1024, 256, 0, "base") RUN(1024, 256, 1, "nt") RUN(1024, 128, 1, "nt")
      RUN(2048, 256, 0, "base") RUN(2048, 256, 1, "nt") RUN(2048, 256, 2, "no-gather") RUN(2048, 256, 3, "nt no-gather") RUN(2048, 256, 5, "nt no-phase2") RUN(2048, 256, 7, "nt no-gather no-ph2")
      RUN(2048, 512, 1, "nt") RUN(4096, 256, 1, "nt") RUN(4096, 512, 1, "nt") RUN(4096, 1024, 1, "nt")
      CK(hipFree(drb));
    }
    Csr A{rows, cols, 0, p, i, nullptr, v};
    double us = time_us([&] { hipLaunchKernelGGL(k_csr_scalar, dim3(8192), dim3(256), 0, 0, A, x, dy); });
    printf("%s scalar row-per-lane            : %7.1f us  %6.0f GB/s\n", name, us, bytes / us / 1e3);
  };
  long long bA = (long long)nnz * 12 + (m + 1) * 4LL + n * 8LL + m * 8LL, bT = (long long)nnz * 12 + (n + 1) * 4LL + m * 8LL + n * 8LL;
  run("A  (m rows, gather n)", m, n, dAp, dAi, dAx, dxm, bA);
  run("At (n rows, gather m)", n, m, dTp, dTi, dTx, dxm, bT);
  return 0;
}
