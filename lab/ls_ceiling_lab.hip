// ls_ceiling_lab.hip -- how far is the LOCKSTEP wave-owned-rows product (scs_amd/csrc/spmv_wave.h, csr_wave_lockstep_kernel<.., 16, 4>) from
// what its own schedule can do with no HBM latency in the CU's memory queue?  (the round-2 ceiling of lab/g4_lab.hip was
// taken on the PLAIN kernel.)  Headline shape: n = 1e6 columns with 10 uniformly random rows each out of m = 2e6; both orientations through the
// library's own layout builder (WaveRowsDev::plan + fill_host).  Modes of a copy of the kernel:
//   0  normal (must time like the library's kernel, which is also run)
//   4  CEILING: the 12 B / entry stream is read from the first two chunks of the unit over and over (L2 / L1 resident) while the gathers keep
//      the real access pattern (uniform in the chunk's share of the column range, which moves upwards with the chunk index, like the
//      bucket-sorted entries do)
//   1  no gather (x index & 63): the stream alone under this schedule
//   6  gathers alone: mode 4's gathers, no stream loads at all (columns from a hash)
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I scs_amd/csrc lab/ls_ceiling_lab.hip -o lab/ls_ceiling_lab ; run: lab/ls_ceiling_lab
#include "spmv_wave.h"
#include <random>
#include <chrono>
using namespace scsamd;

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(16 * 64) void k_ls(WaveView A, const real *__restrict__ x, real *y, int accrows) {
  constexpr int WL_WPB = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char wr_smem[];
  __shared__ int s_nch[WL_WPB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  real *acc = reinterpret_cast<real *>(wr_smem) + (size_t)wave * accrows;
  const unsigned cmask = (1u << A.cbits) - 1;
  const int per_round = gridDim.x * WL_WPB;
  const int nround = (A.nunit + per_round - 1) / per_round;
  for (int rd = 0; rd < nround; ++rd) {
    const int u = rd * per_round + blockIdx.x * WL_WPB + wave;
    const bool live = u < A.nunit;
    int r0 = 0, nr = 0, s = 0, t = 0;
    if (live) { r0 = A.urow[u]; nr = A.urow[u + 1] - r0; s = A.useg[2 * u]; t = A.useg[2 * u + 1]; }
    for (int k = lane; k < nr; k += 64) acc[k] = 0;
    const int nch = (t - s + 255) >> 8;
    if (lane == 0) s_nch[wave] = nch;
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int w = 0; w < WL_WPB; ++w) nmax = s_nch[w] > nmax ? s_nch[w] : nmax;
    const unsigned win = nch > 0 ? (unsigned)A.cols / (unsigned)nch : 1u;
    for (int c = 0; c < nmax; ++c) {
      const bool has = c < nch;
      const int eb = s + c * 256 + lane * 4;
      WrChunk ch;
      ch.w = make_uint4(0, 0, 0, 0);
      ch.v[0] = ch.v[1] = ch.v[2] = ch.v[3] = 1;
      if (has && MODE != 6) ch = wr_load(A, (MODE == 4) ? s + ((c & 1) << 8) + lane * 4 : eb);
      unsigned w[4] = {ch.w.x, ch.w.y, ch.w.z, ch.w.w};
      if (MODE == 4 || MODE == 6) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned h = hash32((unsigned)(eb + i));
          // gather instruction i covers the i-th quarter of the chunk's window (the library's chunk order)
          const unsigned col = (unsigned)c * win + (unsigned)i * (win >> 2) + h % (win >> 2 ? win >> 2 : 1);
          w[i] = (col < (unsigned)A.cols ? col : (unsigned)A.cols - 1) | ((hash32(h) % (unsigned)(nr > 0 ? nr : 1)) << A.cbits);
        }
      }
      real xx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __syncthreads();
        xx[i] = (has && eb + i < t) ? x[MODE == 1 ? ((w[i] & cmask) & 63) : (w[i] & cmask)] : (real)0;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (has && eb + i < t) lds_add(acc + (w[i] >> A.cbits), ch.v[i] * xx[i]);
    }
    for (int k = lane; k < nr; k += 64) y[r0 + k] = acc[k];
    __syncthreads();
  }
}

struct Mat {
  WaveRowsDev w;
  int rows, cols;
};

static void build(Mat &M, int rows, int cols, const std::vector<int> &ptr, const std::vector<int> &idx, const std::vector<real> &val, hipStream_t st) {
  M.rows = rows;
  M.cols = cols;
  M.w.plan(rows, cols, ptr.data());
  std::vector<unsigned> hw;
  std::vector<real> hv;
  long long distinct = 0;
  M.w.fill_host(ptr.data(), idx.data(), val.data(), hw, hv, distinct);
  M.w.alloc_and_upload_plan(st);
  M.w.wrd.upload(hw.data(), M.w.cap, st);
  M.w.val.upload(hv.data(), M.w.cap, st);
  HIP_CHECK(hipStreamSynchronize(st));
  M.w.finish(distinct);
}

template <int MODE>
static double time_mode(Mat &A, Mat &At, DevBuf<real> &xn, DevBuf<real> &ym, DevBuf<real> &xm, DevBuf<real> &yn, hipStream_t st, double *each) {
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ls<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  hipEvent_t e0, e1, e2;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventCreate(&e2));
  double ta = 0, tt = 0;
  const int reps = 30;
  for (int r = -3; r < reps; ++r) {
    HIP_CHECK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_ls<MODE>, dim3(A.w.grid()), dim3(1024), A.w.lds_bytes(), st, A.w.view(), (const real *)xn.p, ym.p, A.w.accrows);
    HIP_CHECK(hipEventRecord(e1, st));
    hipLaunchKernelGGL(k_ls<MODE>, dim3(At.w.grid()), dim3(1024), At.w.lds_bytes(), st, At.w.view(), (const real *)xm.p, yn.p, At.w.accrows);
    HIP_CHECK(hipEventRecord(e2, st));
    HIP_CHECK(hipStreamSynchronize(st));
    float a, b;
    HIP_CHECK(hipEventElapsedTime(&a, e0, e1)); HIP_CHECK(hipEventElapsedTime(&b, e1, e2));
    if (r >= 0) { ta += a; tt += b; }
  }
  each[0] = 1e3 * ta / reps;
  each[1] = 1e3 * tt / reps;
  return each[0] + each[1];
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000000, m = 2 * n, cn = 10;
  hipStream_t st;
  HIP_CHECK(hipStreamCreate(&st));
  std::mt19937 rng(1234);
  std::vector<int> cp(n + 1), ci((size_t)n * cn);
  std::vector<real> cx((size_t)n * cn);
  std::uniform_int_distribution<int> ur(0, m - 1);
  std::normal_distribution<double> nd;
  for (int j = 0; j < n; ++j) {
    cp[j] = j * cn;
    int *r = &ci[(size_t)j * cn];
    for (int k = 0; k < cn; ++k) r[k] = ur(rng);
    std::sort(r, r + cn);
    for (int k = 1; k < cn; ++k) if (r[k] <= r[k - 1]) r[k] = std::min(m - 1, r[k - 1] + 1);
    for (int k = 0; k < cn; ++k) cx[(size_t)j * cn + k] = nd(rng);
  }
  cp[n] = n * cn;
  // CSR(A) by counting sort
  std::vector<int> rp(m + 1, 0), rj(ci.size());
  std::vector<real> rx(ci.size());
  for (int v : ci) rp[v + 1]++;
  for (int i = 0; i < m; ++i) rp[i + 1] += rp[i];
  { std::vector<int> nx(rp.begin(), rp.end() - 1);
    for (int j = 0; j < n; ++j) for (int k = cp[j]; k < cp[j + 1]; ++k) { const int q = nx[ci[k]]++; rj[q] = j; rx[q] = cx[k]; } }
  opt_set("wr_lockstep", "1");
  Mat A, At;
  build(A, m, n, rp, rj, rx, st);   // rows of A gather from an n-vector
  build(At, n, m, cp, ci, cx, st);  // rows of A' gather from an m-vector
  DevBuf<real> xn(n), ym(m), xm(m), yn(n);
  { std::vector<real> h(m, 1.0); xn.upload(h.data(), n, st); xm.upload(h.data(), m, st); HIP_CHECK(hipStreamSynchronize(st)); }
  printf("n=%d m=%d nnz=%d  units %d / %d, grid %d / %d, lines per entry %.3f / %.3f, barriers per chunk %d\n", n, m, n * cn, A.w.nunit, At.w.nunit,
         A.w.grid(), At.w.grid(), A.w.lines_per_entry, At.w.lines_per_entry, A.w.ls_bmode);
  const double bytesA = 12.0 * n * cn + 4.0 * (m + 1) + 8.0 * n + 8.0 * m, bytesT = 12.0 * n * cn + 4.0 * (n + 1) + 8.0 * m + 8.0 * n;
  double e[2];
  auto row = [&](const char *name, double *e) {
    printf("%-78s A %6.1f us (%4.1f%%)  A' %6.1f us (%4.1f%%)  mean %6.1f us => %.3f of 8 TB/s\n", name, e[0], 100 * bytesA / (e[0] * 1e-6) / 8e12, e[1],
           100 * bytesT / (e[1] * 1e-6) / 8e12, 0.5 * (e[0] + e[1]), 0.5 * (bytesA + bytesT) / (0.5 * (e[0] + e[1]) * 1e-6) / 8e12);
  };
  time_mode<0>(A, At, xn, ym, xm, yn, st, e); row("lockstep, normal (matrix from HBM, real gathers)", e);
  time_mode<4>(A, At, xn, ym, xm, yn, st, e); row("lockstep CEILING: stream wrapped into 2 chunks per unit (cache resident), real gather pattern", e);
  time_mode<1>(A, At, xn, ym, xm, yn, st, e); row("lockstep, stream from HBM, no gather (x index & 63)", e);
  time_mode<6>(A, At, xn, ym, xm, yn, st, e); row("lockstep, gathers alone (no stream loads at all)", e);
  time_mode<0>(A, At, xn, ym, xm, yn, st, e); row("lockstep, normal (again: spread)", e);
  return 0;
}
