// lab/pb_lab.hip -- "propagation blocking" experiment for the CG operator (not product code).
//
// Idea under test: the gather-based products are bound by 128-byte L1 line fills per scattered
// 8-byte operand (lab/spmv_lab.hip: 37 us per 1e7 gathers even when L2-resident).  A two-phase
// product never gathers from global memory:
//   producer  a workgroup holds a slice of x in LDS, streams the entries whose column falls in
//             that slice (val 8 B + 16-bit local column), and writes the products val*x[col] --
//             fully coalesced -- to a buffer Q that is ordered for the consumer;
//   consumer  a wave owns RW consecutive rows (accumulators in LDS), streams its contiguous part of Q
//             (product 8 B + 16-bit local row) and adds with ds_add_f64.  No barriers, no sort.
// Both phases are pure streams; the price is 16 B/nnz of product traffic (which the 256 MB
// Infinity Cache may absorb).  mat_vec = P1 (produce A p) -> P2 (consume -> z = R_y^-1 A p in LDS ->
// produce A' z) -> P3 (consume, Gp = R_x p + A' z, partial p.Gp).
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 lab/pb_lab.hip -o lab/pb_lab
// host-only self check of the layout builder (no GPU): lab/pb_lab --hostcheck
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <typename T> T *dev(const std::vector<T> &h) {
  T *p; CK(hipMalloc(&p, (h.size() + 256) * sizeof(T)));
  CK(hipMemset(p, 0, (h.size() + 256) * sizeof(T)));
  CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}
template <typename T> T *devz(size_t n) { T *p; CK(hipMalloc(&p, (n + 256) * sizeof(T))); CK(hipMemset(p, 0, (n + 256) * sizeof(T))); return p; }

// ---------------------------------------------------------------------------------------------
// layout of one product y = M x  (M: rows x cols, given as CSR)
struct HalfHost {
  int rows = 0, cols = 0, CP = 0, RW = 0, nslice = 0, nunit = 0, nruns = 0, nwg = 0;
  size_t nnz = 0;
  std::vector<double> pval;        // producer order: (slice, unit, row, col)
  std::vector<uint16_t> pcol;      // col - slice*CP | 0x8000 on the first entry of a (slice, unit) run
  std::vector<int> gbase;          // per 64 entries: (#run starts before this group) - 1
  std::vector<int> delta;          // per run: consumer start - producer start
  std::vector<int> pwg, pwgslice;  // producer workgroups: entry ranges (nwg + 1) and their slice
  std::vector<uint16_t> crow;      // consumer order: (unit, slice, row, col) -> row - unit*RW
  std::vector<int> cseg;           // nunit + 1
};

// G = producer workgroups per slice
static void build_half(HalfHost &H, int rows, int cols, const std::vector<int> &ptr, const std::vector<int> &idx,
                       const std::vector<double> &val, int CP, int RW, int G) {
  H.rows = rows; H.cols = cols; H.CP = CP; H.RW = RW;
  H.nslice = (cols + CP - 1) / CP; H.nunit = (rows + RW - 1) / RW;
  const size_t nnz = (size_t)ptr[rows]; H.nnz = nnz;
  const int ns = H.nslice, nu = H.nunit;
  std::vector<int> cnt((size_t)nu * ns, 0);
  for (int r = 0; r < rows; ++r) { const int u = r / RW; for (int k = ptr[r]; k < ptr[r + 1]; ++k) cnt[(size_t)u * ns + idx[k] / CP]++; }
  std::vector<int> coff((size_t)nu * ns), poff((size_t)nu * ns); // both indexed [u*ns + j]
  H.cseg.assign(nu + 1, 0);
  { int a = 0; for (int u = 0; u < nu; ++u) { H.cseg[u] = a; for (int j = 0; j < ns; ++j) { coff[(size_t)u * ns + j] = a; a += cnt[(size_t)u * ns + j]; } } H.cseg[nu] = a; }
  std::vector<int> slice_start(ns + 1, 0);
  { int a = 0; for (int j = 0; j < ns; ++j) { slice_start[j] = a; for (int u = 0; u < nu; ++u) { poff[(size_t)u * ns + j] = a; a += cnt[(size_t)u * ns + j]; } } slice_start[ns] = a; }
  const size_t npad = (nnz + 63) / 64 * 64 + 64;
  H.pval.assign(npad, 0.0); H.pcol.assign(npad, 0); H.crow.assign(npad, 0);
  // runs, in producer order
  H.delta.clear();
  std::vector<char> flag(npad, 0);
  for (int j = 0; j < ns; ++j) for (int u = 0; u < nu; ++u) if (cnt[(size_t)u * ns + j]) {
    flag[poff[(size_t)u * ns + j]] = 1; H.delta.push_back(coff[(size_t)u * ns + j] - poff[(size_t)u * ns + j]); }
  H.nruns = (int)H.delta.size();
  H.delta.resize(H.nruns + 128, 0);
  { std::vector<int> pn(poff), cn(coff);
    for (int r = 0; r < rows; ++r) { const int u = r / RW;
      for (int k = ptr[r]; k < ptr[r + 1]; ++k) { const int j = idx[k] / CP; const size_t c = (size_t)u * ns + j;
        const int p = pn[c]++, q = cn[c]++;
        H.pval[p] = val[k]; H.pcol[p] = (uint16_t)(idx[k] - j * CP); H.crow[q] = (uint16_t)(r - u * RW); } } }
  H.gbase.assign(npad / 64, 0);
  { int f = 0; for (size_t e = 0; e < npad; ++e) { if ((e & 63) == 0) H.gbase[e >> 6] = f - 1; if (flag[e]) { H.pcol[e] |= 0x8000; ++f; } } }
  // producer workgroups: every slice cut into G ranges at multiples of 64 entries
  H.pwg.clear(); H.pwgslice.clear();
  for (int j = 0; j < ns; ++j) { const int a = slice_start[j], b = slice_start[j + 1]; if (a == b) continue;
    const int len = b - a; int step = ((len + G - 1) / G + 63) / 64 * 64; if (step < 64) step = 64;
    for (int s = a; s < b; s += step) { H.pwg.push_back(s); H.pwgslice.push_back(j); } }
  H.nwg = (int)H.pwgslice.size();
  // ranges end where the next begins, except at slice ends: store explicit end list instead
  // (pwg holds starts; ends computed here)
  std::vector<int> ends(H.nwg);
  for (int w = 0; w < H.nwg; ++w) { const int j = H.pwgslice[w]; ends[w] = (w + 1 < H.nwg && H.pwgslice[w + 1] == j) ? H.pwg[w + 1] : slice_start[j + 1]; }
  H.pwg.insert(H.pwg.end(), ends.begin(), ends.end()); // [0,nwg) starts, [nwg,2nwg) ends
}

// host emulation of producer + consumer (for --hostcheck)
static void host_apply(const HalfHost &H, const std::vector<double> &x, std::vector<double> &y) {
  std::vector<double> Q(H.pval.size(), 0.0);
  for (int w = 0; w < H.nwg; ++w) { const int j = H.pwgslice[w], a = H.pwg[w], b = H.pwg[H.nwg + w];
    for (int g = a >> 6; g < (b + 63) >> 6; ++g) { int R = H.gbase[g];
      for (int l = 0; l < 64; ++l) { const int e = g * 64 + l; const uint16_t c = H.pcol[e]; if (c & 0x8000) ++R;
        if (e >= a && e < b) Q[(size_t)H.delta[R] + e] = H.pval[e] * x[(size_t)j * H.CP + (c & 0x7fff)]; } } }
  y.assign(H.rows, 0.0);
  for (int u = 0; u < H.nunit; ++u) for (int e = H.cseg[u]; e < H.cseg[u + 1]; ++e) y[(size_t)u * H.RW + H.crow[e]] += Q[e];
}

struct Half { // device view
  int rows, cols, CP, RW, nslice, nunit, nwg;
  const double *pval; const uint16_t *pcol; const int *gbase, *delta, *pwg, *pwgslice; const uint16_t *crow; const int *cseg;
};
static Half upload(const HalfHost &H) {
  return Half{H.rows, H.cols, H.CP, H.RW, H.nslice, H.nunit, H.nwg, dev(H.pval), dev(H.pcol), dev(H.gbase), dev(H.delta), dev(H.pwg), dev(H.pwgslice), dev(H.crow), dev(H.cseg)};
}

// ---------------------------------------------------------------------------------------------
#define DELTA_LDS 4096
// producer body: xs = slice of the multiplied vector in LDS.  entries [a,b) of the producer order.
template <int U, int NT, int NW, int MODE = 0> __device__ __forceinline__ void produce(const Half &H, const double *xs, int a, int b, double *__restrict__ Q, int *dl) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g0 = a >> 6, g1 = (b + 63) >> 6;
  // stage this range's run table
  const int rlo = max(H.gbase[g0], 0);
  const int rhi = H.gbase[g1 - 1] + 64; // inclusive upper bound on run ids used here
  const bool lds_delta = (rhi - rlo + 1) <= DELTA_LDS;
  if (lds_delta) for (int k = tid; k <= rhi - rlo; k += NW * 64) dl[k] = H.delta[rlo + k];
  __syncthreads();
  const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1);
  for (int g = g0 + wave; g < g1; g += NW * U) {
    uint16_t c[U]; double v[U]; int gb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int gg = g + u * NW; const bool ok = gg < g1; const int e = gg * 64 + lane;
      c[u] = ok ? (NT ? __builtin_nontemporal_load(H.pcol + e) : H.pcol[e]) : (uint16_t)0;
      v[u] = ok ? (NT ? __builtin_nontemporal_load(H.pval + e) : H.pval[e]) : 0.0;
      gb[u] = ok ? H.gbase[gg] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int gg = g + u * NW; const int e = gg * 64 + lane;
      const unsigned long long ball = __ballot(c[u] & 0x8000);
      const int R = gb[u] + __popcll(ball & le);
      const bool in = gg < g1 && e >= a && e < b;
      if (in) {
        const int d = lds_delta ? dl[R - rlo] : H.delta[R];
        const double q = v[u] * xs[c[u] & 0x7fff];
        if (MODE == 1) { if (q == 1.2345e300) Q[e] = q; }
        else if (MODE == 2) Q[e] = q;
        else if (NT) __builtin_nontemporal_store(q, Q + (size_t)d + e); else Q[(size_t)d + e] = q;
      }
    }
  }
}

template <int U, int NT, int NW, int MODE = 0> __global__ __launch_bounds__(NW * 64) void k_prod(Half H, const double *__restrict__ x, double *__restrict__ Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *xs = reinterpret_cast<double *>(smem);
  int *dl = reinterpret_cast<int *>(xs + H.CP);
  const int w = blockIdx.x, j = H.pwgslice[w], a = H.pwg[w], b = H.pwg[H.nwg + w];
  const int c0 = j * H.CP, nc = min(H.CP, H.cols - c0);
  for (int k = threadIdx.x; k < nc; k += NW * 64) xs[k] = x[c0 + k];
  produce<U, NT, NW, MODE>(H, xs, a, b, Q, dl); // begins with a barrier
}

// consumer body for one wave: unit u, accumulators a[RW] (LDS, zeroed here)
template <int U, int NT, int MODE = 0> __device__ __forceinline__ void consume(const Half &H, int u, const double *__restrict__ Q, double *acc) {
  const int lane = threadIdx.x & 63;
  for (int k = lane; k < H.RW; k += 64) acc[k] = 0.0;
  if (u >= H.nunit) return;
  const int s = H.cseg[u], t = H.cseg[u + 1];
  for (int e0 = s; e0 < t; e0 += 64 * U) {
    double q[U]; uint16_t r[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { const int e = e0 + i * 64 + lane; const bool ok = e < t;
      q[i] = ok ? (NT ? __builtin_nontemporal_load(Q + e) : Q[e]) : 0.0;
      r[i] = ok ? (NT ? __builtin_nontemporal_load(H.crow + e) : H.crow[e]) : (uint16_t)0; }
#pragma unroll
    for (int i = 0; i < U; ++i) { const int e = e0 + i * 64 + lane;
      if (MODE == 1) { if (e < t && q[i] == 1.2345e300) acc[r[i]] = q[i]; }
      else if (e < t) __hip_atomic_fetch_add(acc + r[i], q[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  }
}

// plain consumer: y = acc (EPI 0), y = acc / d (EPI 1), y = acc + d*xin, partial dot (EPI 2)
template <int U, int NT, int WPB, int EPI, int MODE = 0> __global__ __launch_bounds__(WPB * 64) void k_cons(Half H, const double *__restrict__ Q, double *__restrict__ y,
                                                                                     const double *__restrict__ d, const double *__restrict__ xin, double *partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *accall = reinterpret_cast<double *>(smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = blockIdx.x * WPB + wave;
  double *acc = accall + wave * H.RW;
  consume<U, NT, MODE>(H, u, Q, acc);
  double dot = 0;
  if (u < H.nunit) {
    const int r0 = u * H.RW, nr = min(H.RW, H.rows - r0);
    for (int k = lane; k < nr; k += 64) {
      double o = acc[k];
      if (EPI == 1) o = o / d[r0 + k];
      if (EPI == 2) { const double xr = xin[r0 + k]; o = o + d[r0 + k] * xr; dot += xr * o; }
      y[r0 + k] = o;
    }
  }
  if (EPI == 2) {
    __shared__ double red[WPB];
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int i = 0; i < WPB; ++i) s += red[i]; partial[blockIdx.x] = s; }
  }
}

// fused middle kernel: consume units of block i (WPB waves * RW rows == CP of the next half), z = acc / ry stays
// in LDS, then produce A' z for slice i.
template <int U, int NT, int WPB> __global__ __launch_bounds__(WPB * 64) void k_mid(Half H1, Half H2, const double *__restrict__ Q1, double *__restrict__ Q2,
                                                                                const double *__restrict__ ry, double *zout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *accall = reinterpret_cast<double *>(smem);
  int *dl = reinterpret_cast<int *>(accall + WPB * H1.RW);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = blockIdx.x;             // producer workgroup of H2 == slice w of H2 (G = 1) == row block w of H1
  const int u = w * WPB + wave;
  double *acc = accall + wave * H1.RW;
  consume<U, NT>(H1, u, Q1, acc);
  if (u < H1.nunit) {
    const int r0 = u * H1.RW, nr = min(H1.RW, H1.rows - r0);
    for (int k = lane; k < nr; k += 64) { const double z = acc[k] / ry[r0 + k]; acc[k] = z; if (zout) zout[r0 + k] = z; }
  }
  const int a = H2.pwg[w], b = H2.pwg[H2.nwg + w];
  produce<U, NT, WPB>(H2, accall, a, b, Q2, dl); // begins with a barrier
}

// reference: scalar CSR
__global__ void k_csr_scalar(int rows, const int *ptr, const int *idx, const double *val, const double *__restrict__ x, double *y) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    double acc = 0; for (int k = ptr[r]; k < ptr[r + 1]; ++k) acc += val[k] * x[idx[k]]; y[r] = acc; }
}
__global__ void k_divide(int n, double *y, const double *d) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] /= d[i]; }
__global__ void k_axpy_gp(int n, double *y, const double *d, const double *x) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] += d[i] * x[i]; }
// MALL probes
__global__ void k_fill(double2 *b, size_t n2, double v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) b[i] = double2{v, v}; }
__global__ void k_read(const double2 *__restrict__ a, size_t n2, double *out) { double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { double2 t = a[i]; s += t.x + t.y; } if (s == 1.2345) out[0] = s; }

struct Ev { hipEvent_t e; Ev() { CK(hipEventCreate(&e)); } };
static double el(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return 1e3 * ms; }

static void gen(int n, int m, int cn, std::vector<int> &Tp, std::vector<int> &Ti, std::vector<double> &Tx, std::vector<int> &Ap, std::vector<int> &Ai, std::vector<double> &Ax) {
  size_t nnz = (size_t)n * cn; std::mt19937_64 rng(1);
  Tp.resize(n + 1); Ti.resize(nnz); Tx.resize(nnz);
  for (int j = 0; j < n; ++j) { Tp[j] = j * cn; int *r = &Ti[(size_t)j * cn];
    for (;;) { for (int k = 0; k < cn; ++k) r[k] = (int)(rng() % m); std::sort(r, r + cn); if (std::adjacent_find(r, r + cn) == r + cn) break; }
    for (int k = 0; k < cn; ++k) Tx[(size_t)j * cn + k] = (double)(rng() % 2001) / 1000.0 - 1.0; }
  Tp[n] = (int)nnz;
  Ap.assign(m + 1, 0); Ai.resize(nnz); Ax.resize(nnz);
  for (size_t k = 0; k < nnz; ++k) Ap[Ti[k] + 1]++;
  for (int i = 0; i < m; ++i) Ap[i + 1] += Ap[i];
  std::vector<int> nx(Ap.begin(), Ap.end() - 1);
  for (int j = 0; j < n; ++j) for (int k = Tp[j]; k < Tp[j + 1]; ++k) { int q = nx[Ti[k]]++; Ai[q] = j; Ax[q] = Tx[k]; }
}

template <int U, int NT, int WPB> static void run_config(const char *tag, int n, int m, const HalfHost &h1, const HalfHost &h2, const Half &H1, const Half &H2,
                                                         const double *dp, const double *dry, const double *drx, double *dgp, const double *dgp_ref, double *dz, double *Q1, double *Q2, double *dpart,
                                                         int reps, double *dbig, size_t bigN) {
  const int RW1 = h1.RW, RW2 = h2.RW;
  const size_t lds_p1 = (size_t)h1.CP * 8 + DELTA_LDS * 4;
  const size_t lds_mid = (size_t)WPB * RW1 * 8 + DELTA_LDS * 4;
  constexpr int WPB3 = 4;
  const size_t lds_p3 = (size_t)WPB3 * RW2 * 8;
  const int g1 = H1.nwg, g2 = (h1.nunit + WPB - 1) / WPB, g3 = (h2.nunit + WPB3 - 1) / WPB3;
  if (g2 != H2.nwg) { printf("%s: block mismatch g2=%d nwg2=%d\n", tag, g2, H2.nwg); return; }
  auto p1 = [&] { hipLaunchKernelGGL((k_prod<U, NT, 8>), dim3(g1), dim3(512), lds_p1, 0, H1, dp, Q1); };
  auto p2 = [&] { hipLaunchKernelGGL((k_mid<U, NT, WPB>), dim3(g2), dim3(WPB * 64), lds_mid, 0, H1, H2, Q1, Q2, dry, (double *)nullptr); };
  auto p3 = [&] { hipLaunchKernelGGL((k_cons<U, NT, WPB3, 2>), dim3(g3), dim3(WPB3 * 64), lds_p3, 0, H2, Q2, dgp, drx, dp, dpart); };
  CK(hipFuncSetAttribute((const void *)k_prod<U, NT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p1));
  CK(hipFuncSetAttribute((const void *)k_mid<U, NT, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mid));
  CK(hipFuncSetAttribute((const void *)k_cons<U, NT, WPB3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p3));
  CK(hipMemset(dgp, 0, (size_t)n * 8));
  p1(); p2(); p3(); CK(hipDeviceSynchronize());
  std::vector<double> a(n), b(n); CK(hipMemcpy(a.data(), dgp, (size_t)n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dgp_ref, (size_t)n * 8, hipMemcpyDeviceToHost));
  double err = 0, nb = 0; for (int i = 0; i < n; ++i) { err = std::max(err, fabs(a[i] - b[i])); nb = std::max(nb, fabs(b[i])); }
  // bitwise reproducibility
  p1(); p2(); p3(); CK(hipDeviceSynchronize());
  std::vector<double> a2(n); CK(hipMemcpy(a2.data(), dgp, (size_t)n * 8, hipMemcpyDeviceToHost));
  const bool same = memcmp(a.data(), a2.data(), (size_t)n * 8) == 0;
  Ev e0, e1, e2, e3;
  double t1 = 0, t2 = 0, t3 = 0, tall = 0;
  for (int i = 0; i < 3; ++i) { p1(); p2(); p3(); }
  CK(hipDeviceSynchronize());
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0.e)); p1(); CK(hipEventRecord(e1.e)); p2(); CK(hipEventRecord(e2.e)); p3(); CK(hipEventRecord(e3.e)); CK(hipEventSynchronize(e3.e));
    t1 += el(e0.e, e1.e); t2 += el(e1.e, e2.e); t3 += el(e2.e, e3.e); }
  CK(hipEventRecord(e0.e)); for (int i = 0; i < reps; ++i) { p1(); p2(); p3(); } CK(hipEventRecord(e3.e)); CK(hipEventSynchronize(e3.e)); tall = el(e0.e, e3.e) / reps;
  // same, with a 256 MB stream in between every mat_vec (evicts the Infinity Cache between iterations, like the rest of a CG iteration partly does)
  double tcold = 0;
  for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const double2 *)dbig, bigN / 2, dpart + 8000);
    CK(hipEventRecord(e0.e)); p1(); p2(); p3(); CK(hipEventRecord(e3.e)); CK(hipEventSynchronize(e3.e)); tcold += el(e0.e, e3.e); }
  const double nnz = (double)h1.nnz;
  printf("%-34s err %.2e (|Gp| %.2e) bitwise-repeatable %d | P1 %.1f  P2 %.1f  P3 %.1f us | back-to-back %.1f us/mat_vec  after-256MB-flush %.1f | 300MB/mat_vec => %.2f TB/s (%.1f%% of 8)\n",
         tag, err, nb, (int)same, t1 / reps, t2 / reps, t3 / reps, tall, tcold / reps, 300e6 * (nnz / 1e7) / tall / 1e6, 300e6 * (nnz / 1e7) / tall / 1e6 / 8 * 100);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const bool hostcheck = argc > 1 && !strcmp(argv[1], "--hostcheck");
  int n = hostcheck ? 20000 : (argc > 1 ? atoi(argv[1]) : 1000000); int cn = 10; int m = 2 * n;
  std::vector<int> Tp, Ti, Ap, Ai; std::vector<double> Tx, Ax;
  gen(n, m, cn, Tp, Ti, Tx, Ap, Ai, Ax);
  const size_t nnz = Ti.size();
  std::mt19937_64 rng(7);
  std::vector<double> hp(n), hry(m), hrx(n);
  for (auto &v : hp) v = (double)(rng() % 2001) / 1000.0 - 1.0;
  for (auto &v : hry) v = 0.5 + (double)(rng() % 1000) / 1000.0;
  for (auto &v : hrx) v = 0.1 + (double)(rng() % 1000) / 10000.0;
  if (hostcheck) {
    for (int RW : {512, 1024}) for (int CP : {4096, 8192}) for (int G : {1, 3}) {
      HalfHost h1, h2; build_half(h1, m, n, Ap, Ai, Ax, CP, RW, G); build_half(h2, n, m, Tp, Ti, Tx, CP, RW, G);
      std::vector<double> y1, y2, r1(m, 0.0), r2(n, 0.0), xm(m);
      for (int i = 0; i < m; ++i) xm[i] = hry[i];
      host_apply(h1, hp, y1); host_apply(h2, xm, y2);
      for (int r = 0; r < m; ++r) for (int k = Ap[r]; k < Ap[r + 1]; ++k) r1[r] += Ax[k] * hp[Ai[k]];
      for (int r = 0; r < n; ++r) for (int k = Tp[r]; k < Tp[r + 1]; ++k) r2[r] += Tx[k] * xm[Ti[k]];
      double e1 = 0, e2 = 0; for (int i = 0; i < m; ++i) e1 = std::max(e1, fabs(y1[i] - r1[i])); for (int i = 0; i < n; ++i) e2 = std::max(e2, fabs(y2[i] - r2[i]));
      printf("hostcheck RW=%d CP=%d G=%d: A err %.2e (runs %d, wgs %d)  At err %.2e (runs %d, wgs %d)\n", RW, CP, G, e1, h1.nruns, h1.nwg, e2, h2.nruns, h2.nwg);
    }
    return 0;
  }
  printf("n=%d m=%d nnz=%zu\n", n, m, nnz);
  int *dAp = dev(Ap), *dAi = dev(Ai), *dTp = dev(Tp), *dTi = dev(Ti);
  double *dAx = dev(Ax), *dTx = dev(Tx), *dp = dev(hp), *dry = dev(hry), *drx = dev(hrx);
  double *dz = devz<double>(m), *dgp_ref = devz<double>(n), *dgp = devz<double>(n), *dpart = devz<double>(16384);
  double *Q1 = devz<double>(nnz + 4096), *Q2 = devz<double>(nnz + 4096);
  const size_t bigN = 32u << 20; // 256 MB of doubles
  double *dbig = devz<double>(bigN);
  // reference Gp = rx.*p + A' (ry^-1 (A p))
  hipLaunchKernelGGL(k_csr_scalar, dim3(8192), dim3(256), 0, 0, m, dAp, dAi, dAx, dp, dz);
  hipLaunchKernelGGL(k_divide, dim3(4096), dim3(256), 0, 0, m, dz, dry);
  hipLaunchKernelGGL(k_csr_scalar, dim3(8192), dim3(256), 0, 0, n, dTp, dTi, dTx, dz, dgp_ref);
  hipLaunchKernelGGL(k_axpy_gp, dim3(4096), dim3(256), 0, 0, n, dgp_ref, drx, dp);
  CK(hipDeviceSynchronize());

  // MALL probes: read 80 MB right after writing it vs after a 256 MB flush
  { Ev a, b; const size_t n2 = nnz / 2; double tw = 0, tr_hot = 0, tr_cold = 0, tbig = 0; const int reps = 10;
    for (int i = 0; i < reps + 2; ++i) {
      CK(hipEventRecord(a.e)); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double2 *)Q1, n2, 1.0); CK(hipEventRecord(b.e)); CK(hipEventSynchronize(b.e)); if (i >= 2) tw += el(a.e, b.e);
      CK(hipEventRecord(a.e)); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const double2 *)Q1, n2, dpart); CK(hipEventRecord(b.e)); CK(hipEventSynchronize(b.e)); if (i >= 2) tr_hot += el(a.e, b.e);
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double2 *)Q1, n2, 2.0);
      CK(hipEventRecord(a.e)); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const double2 *)dbig, bigN / 2, dpart); CK(hipEventRecord(b.e)); CK(hipEventSynchronize(b.e)); if (i >= 2) tbig += el(a.e, b.e);
      CK(hipEventRecord(a.e)); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const double2 *)Q1, n2, dpart); CK(hipEventRecord(b.e)); CK(hipEventSynchronize(b.e)); if (i >= 2) tr_cold += el(a.e, b.e);
    }
    printf("MALL probe (80 MB): write %.1f us (%.2f TB/s) | read right after write %.1f us (%.2f TB/s) | read after 256 MB flush %.1f us (%.2f TB/s) | 256 MB stream read %.1f us (%.2f TB/s)\n",
           tw / reps, 80e6 / (tw / reps) / 1e6, tr_hot / reps, 80e6 / (tr_hot / reps) / 1e6, tr_cold / reps, 80e6 / (tr_cold / reps) / 1e6, tbig / reps, 268.4e6 / (tbig / reps) / 1e6); }


  if (getenv("PB_DIAG")) {
    HalfHost h1, h2; build_half(h1, m, n, Ap, Ai, Ax, 8192, 512, 4); build_half(h2, n, m, Tp, Ti, Tx, 8192, 512, 4);
    Half H1 = upload(h1), H2 = upload(h2);
    auto T = [&](auto f) { Ev a, b; for (int i = 0; i < 3; ++i) f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a.e)); for (int i = 0; i < 20; ++i) f(); CK(hipEventRecord(b.e)); CK(hipEventSynchronize(b.e)); return el(a.e, b.e) / 20; };
    const size_t lp = (size_t)8192 * 8 + DELTA_LDS * 4;
#define PD(U, NT, MODE, HH, X) { CK(hipFuncSetAttribute((const void *)k_prod<U, NT, 8, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp)); \
      double t = T([&] { hipLaunchKernelGGL((k_prod<U, NT, 8, MODE>), dim3(HH.nwg), dim3(512), lp, 0, HH, X, Q1); }); printf("  prod U%d NT%d MODE%d (%s, %d wgs): %.1f us\n", U, NT, MODE, #HH, HH.nwg, t); }
    PD(8, 0, 0, H1, dp) PD(8, 0, 1, H1, dp) PD(8, 0, 2, H1, dp) PD(4, 0, 2, H1, dp) PD(8, 1, 2, H1, dp)
    PD(8, 0, 0, H2, dz) PD(8, 0, 1, H2, dz) PD(8, 0, 2, H2, dz)
#define CD(U, NT, WPB, MODE, HH) { const size_t lc = (size_t)WPB * 512 * 8; int g = (HH.nunit + WPB - 1) / WPB; \
      double t = T([&] { hipLaunchKernelGGL((k_cons<U, NT, WPB, 0, MODE>), dim3(g), dim3(WPB * 64), lc, 0, HH, Q1, dgp == nullptr ? dgp : dz, dry, dp, dpart); }); printf("  cons U%d NT%d WPB%d MODE%d (%s, %d wgs): %.1f us\n", U, NT, WPB, MODE, #HH, g, t); }
    CD(8, 0, 4, 0, H1) CD(8, 0, 4, 1, H1) CD(8, 1, 4, 0, H1) CD(8, 0, 8, 0, H1) CD(8, 0, 16, 0, H1) CD(4, 0, 4, 0, H1) CD(8, 0, 2, 0, H1)
    CD(8, 0, 4, 0, H2) CD(8, 0, 4, 1, H2) CD(8, 1, 4, 0, H2)
    fflush(stdout);
  }

  const int reps = 20;
  struct Cfg { int CP, RW, G; };
  for (Cfg c : {Cfg{8192, 1024, 4}, Cfg{8192, 512, 4}, Cfg{4096, 512, 2}, Cfg{4096, 1024, 2}}) {
    HalfHost h1, h2;
    build_half(h1, m, n, Ap, Ai, Ax, c.CP, c.RW, c.G);   // z = A p   (slices of p, units of rows of A)
    build_half(h2, n, m, Tp, Ti, Tx, c.CP, c.RW, 1);     // Gp = A' z (slices of z == row blocks of A, G = 1)
    Half H1 = upload(h1), H2 = upload(h2);
    printf("-- CP=%d RW=%d G=%d: P1 wgs %d, runs %d (avg %.1f entries); mid wgs %d, runs %d (avg %.1f); units %d / %d\n", c.CP, c.RW, c.G, h1.nwg, h1.nruns, (double)nnz / h1.nruns, h2.nwg, h2.nruns, (double)nnz / h2.nruns, h1.nunit, h2.nunit);
    char tag[128];
#define RUNCFG(U, NT, WPB) if (c.CP == WPB * c.RW) { snprintf(tag, sizeof tag, "CP%d RW%d G%d U%d NT%d WPB%d", c.CP, c.RW, c.G, U, NT, WPB); \
      run_config<U, NT, WPB>(tag, n, m, h1, h2, H1, H2, dp, dry, drx, dgp, dgp_ref, dz, Q1, Q2, dpart, reps, dbig, bigN); }
    RUNCFG(4, 0, 8) RUNCFG(8, 0, 8) RUNCFG(4, 1, 8) RUNCFG(8, 1, 8) RUNCFG(4, 0, 16) RUNCFG(8, 0, 16) RUNCFG(4, 0, 4) RUNCFG(8, 0, 4) RUNCFG(8, 1, 4) RUNCFG(8, 1, 16)
    for (const void *p : {(const void *)H1.pval, (const void *)H1.pcol, (const void *)H1.gbase, (const void *)H1.delta, (const void *)H1.pwg, (const void *)H1.pwgslice, (const void *)H1.crow, (const void *)H1.cseg,
                          (const void *)H2.pval, (const void *)H2.pcol, (const void *)H2.gbase, (const void *)H2.delta, (const void *)H2.pwg, (const void *)H2.pwgslice, (const void *)H2.crow, (const void *)H2.cseg}) CK(hipFree((void *)p));
  }
  return 0;
}
