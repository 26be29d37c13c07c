// lab/sgather_lab.hip -- can the SCALAR memory path (s_load through the scalar data cache) carry part of the SpMV's gathers?
// The product is bound by the vector L1's miss handling (one line fill per gather, profiles/r2_g4_lab.md); the scalar cache is a separate
// path to L2 with 64-byte lines.  Each wave takes 64 column indices (coalesced vector load), and fetches x[col] for S of its lanes with
// s_load_dwordx2 (address from v_readlane) and for the other 64 - S lanes with the usual vector gather.  Not product code.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 lab/sgather_lab.hip -o lab/sgather_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

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

// S lanes (the first S of every wave) through the scalar path, in batches of 8 outstanding s_loads
template <int S> __global__ __launch_bounds__(256) void k_mix(const int *__restrict__ idx, const double *__restrict__ x, double *out, size_t nnz) {
  double acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  for (size_t i0 = blockIdx.x * (size_t)blockDim.x; i0 < nnz; i0 += stride) { // (uniform bounds: readlane needs every lane's index valid)
    const size_t i = i0 + threadIdx.x;
    const int c = i < nnz ? idx[i] : 0;
    double v = 0;
    if (lane >= S) v = x[c];
    if (S > 0) {
      double sv = 0;
#pragma unroll
      for (int b = 0; b < S; b += 8) {
        double r[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
          const int cc = __builtin_amdgcn_readlane(c, b + l);
          const double *p = x + cc;
          asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(r[l]) : "s"(p));
        }
        // the loads are asynchronous: every use of r[] must come after this wait, and no register of r[] may be reused before it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]), "+s"(r[4]), "+s"(r[5]), "+s"(r[6]), "+s"(r[7]) : : "memory");
#pragma unroll
        for (int l = 0; l < 8; ++l) sv = lane == b + l ? r[l] : sv;
      }
      if (lane < S) v = sv;
    }
    acc += v;
  }
  if (acc == 12345.678) out[0] = acc;
}

int main(int argc, char **argv) {
  const size_t nnz = 10000000;
  const int tab = argc > 1 ? atoi(argv[1]) : 1000000;
  std::vector<int> h(nnz);
  std::mt19937 g(1);
  for (auto &v : h) v = (int)(g() % (unsigned)tab);
  int *idx; CK(hipMalloc(&idx, nnz * sizeof(int))); CK(hipMemcpy(idx, h.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
  double *x; CK(hipMalloc(&x, (size_t)tab * 8)); CK(hipMemset(x, 0, (size_t)tab * 8));
  double *out; CK(hipMalloc(&out, 64));
  const int grid = 256 * 8;
  printf("table %d doubles, 1e7 gathers, grid %d x 256\n", tab, grid);
#define RUN(S) printf("  %2d of 64 lanes through the scalar path: %.1f us\n", S, time_us([&] { hipLaunchKernelGGL(k_mix<S>, dim3(grid), dim3(256), 0, 0, idx, x, out, nnz); }));
  RUN(0) RUN(8) RUN(16) RUN(32) RUN(64)
  return 0;
}
