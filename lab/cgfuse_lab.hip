// lab/cgfuse_lab.hip -- VERDICT r2 item 3: does fusing the two vector kernels of a PCG iteration
// (k_cg_update: alpha, x += alpha p, r -= alpha Gp, z = M r, partials of z'r and |r|;  k_cg_direction: stop test,
// beta, p = z + beta p) behind ONE grid barrier pay?  Not product code.
//
// What the fusion saves: p and z stay in registers across the barrier (z is never written, z and p are not re-read:
// 24 of 88 MB at n = 1e6) and one kernel boundary; what it costs: the grid barrier (guide price list: 4.1 / 5.9 us
// at 256 / 512 workgroups) and a co-residency requirement.
//
// Sequence timed (what a CG iteration looks like to the caches): two streaming kernels of 150 MB each (stand-ins for
// the two products: they flush the vectors out of L2 and the Infinity Cache exactly as the matrix streams do), then
// either [update][direction] or [fused].  HIP events around `reps` iterations; the stand-in time is measured alone
// and subtracted.
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 lab/cgfuse_lab.hip -o lab/cgfuse_lab ;  lab/cgfuse_lab [n] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int BLK = 256;
struct alignas(16) d2 { double v[2]; };
__device__ __forceinline__ d2 ld2(const double *p, size_t i) { return reinterpret_cast<const d2 *>(p)[i]; }
__device__ __forceinline__ void st2(double *p, size_t i, const d2 &x) { reinterpret_cast<d2 *>(p)[i] = x; }
template <typename T> __device__ __forceinline__ T wsum(T v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }
template <typename T> __device__ __forceinline__ T wmax(T v) { for (int o = 32; o > 0; o >>= 1) { T w = __shfl_down(v, o, 64); v = w > v ? w : v; } return v; }
__device__ __forceinline__ double bsum(double v, double *sh) {
  v = wsum(v); const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads(); if (l == 0) sh[w] = v; __syncthreads();
  double s = sh[0]; for (int i = 1; i < nw; ++i) s += sh[i]; return s;
}
__device__ __forceinline__ double bmax(double v, double *sh) {
  v = wmax(v); const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads(); if (l == 0) sh[w] = v; __syncthreads();
  double s = sh[0]; for (int i = 1; i < nw; ++i) s = sh[i] > s ? sh[i] : s; return s;
}

struct Ctl { double ztr[2]; double norm_r, tol; int done, iters; unsigned epoch, fault; };
// two-level arrival: 8 group counters (blockIdx % 8: the dispatcher is observed to place block b on XCD b % 8, used for
// speed only), a top counter, a generation word everybody polls.  Counters are monotonic: barrier number E (1-based,
// read from ctl->epoch at kernel start) is complete when gen == E.
struct Bar { unsigned grp[8 * 32]; unsigned top[32]; unsigned gen[32]; };

__global__ void k_stream(const double *__restrict__ a, size_t n2, double *sink) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { const d2 v = ld2(a, i); s += v.v[0] + v.v[1]; }
  if (s == 1.2345e300) sink[0] = s;
}

__global__ __launch_bounds__(BLK) void k_update(double *x, double *r, double *z, const double *__restrict__ p, const double *__restrict__ Gp,
                                                const double *__restrict__ M, int n, const double *ppgp, int cnt, double *pztr, double *pmax,
                                                const Ctl *ctl, int par) {
  __shared__ double red[4];
  const int gtid = blockIdx.x * BLK + threadIdx.x, gs = gridDim.x * BLK, nv = n / 2;
  double ps = 0;
  for (int i = threadIdx.x; i < cnt; i += BLK) ps += ppgp[i];
  if (ctl->done) return;
  const double alpha = ctl->ztr[par] / bsum(ps, red);
  double ztr = 0, mx = 0;
  for (int iv = gtid; iv < nv; iv += gs) {
    const d2 P = ld2(p, iv), G = ld2(Gp, iv), Mv = ld2(M, iv);
    d2 X = ld2(x, iv), R = ld2(r, iv), Z;
    for (int e = 0; e < 2; ++e) {
      X.v[e] += alpha * P.v[e];
      const double ri = R.v[e] + (-alpha) * G.v[e];
      R.v[e] = ri; const double zi = ri * Mv.v[e]; Z.v[e] = zi; ztr += zi * ri;
      const double a = fabs(ri); mx = a > mx ? a : mx;
    }
    st2(x, iv, X); st2(r, iv, R); st2(z, iv, Z);
  }
  ztr = bsum(ztr, red); mx = bmax(mx, red);
  if (threadIdx.x == 0) { pztr[blockIdx.x] = ztr; pmax[blockIdx.x] = mx; }
}
__global__ __launch_bounds__(BLK) void k_direction(double *p, const double *__restrict__ z, int n, const double *pztr, const double *pmax, int pc, Ctl *ctl, int par) {
  __shared__ double red[4];
  const int gtid = blockIdx.x * BLK + threadIdx.x, gs = gridDim.x * BLK, nv = n / 2;
  double zs = 0, ms = 0;
  for (int i = threadIdx.x; i < pc; i += BLK) { zs += pztr[i]; const double v = pmax[i]; ms = v > ms ? v : ms; }
  if (ctl->done) return;
  const double ztr = bsum(zs, red), nr = bmax(ms, red), zp = ctl->ztr[par];
  const bool conv = nr < ctl->tol;
  if (!conv) {
    const double beta = ztr / zp;
    for (int iv = gtid; iv < nv; iv += gs) { const d2 Z = ld2(z, iv); d2 P = ld2(p, iv); for (int e = 0; e < 2; ++e) P.v[e] = Z.v[e] + beta * P.v[e]; st2(p, iv, P); }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->ztr[par ^ 1] = ztr; ctl->norm_r = nr; ctl->iters += 1; }
}

__device__ __forceinline__ bool grid_barrier(Bar *bar, unsigned E, Ctl *ctl) {
  __shared__ int ok_sh;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 7, members = (gridDim.x - g + 7) / 8, ngrp = gridDim.x < 8 ? gridDim.x : 8;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned a = __hip_atomic_fetch_add(&bar->grp[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == E * members) {
      const unsigned t = __hip_atomic_fetch_add(&bar->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1 == E * ngrp) __hip_atomic_store(&bar->gen[0], E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int ok = 0;
    for (unsigned spin = 0; spin < (1u << 22); ++spin) {
      if (__hip_atomic_load(&bar->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= E) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) ctl->fault = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok_sh = ok;
  }
  __syncthreads();
  return ok_sh != 0;
}

// CH = vector chunks per lane kept in registers across the barrier
template <int CH>
__global__ __launch_bounds__(BLK) void k_updir(double *x, double *r, double *p, const double *__restrict__ Gp, const double *__restrict__ M, int n,
                                               const double *ppgp, int cnt, double *pztr, double *pmax, Ctl *ctl, Bar *bar, int par) {
  __shared__ double red[4];
  const int gtid = blockIdx.x * BLK + threadIdx.x, gs = gridDim.x * BLK, nv = n / 2;
  const unsigned E = ctl->epoch + 1;
  double ps = 0;
  for (int i = threadIdx.x; i < cnt; i += BLK) ps += ppgp[i];
  if (ctl->done) return;
  const double zp = ctl->ztr[par], tol = ctl->tol;
  const double alpha = zp / bsum(ps, red);
  double ztr = 0, mx = 0;
  d2 Pk[CH], Zk[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int iv = gtid + c * gs;
    if (iv < nv) {
      const d2 P = ld2(p, iv), G = ld2(Gp, iv), Mv = ld2(M, iv);
      d2 X = ld2(x, iv), R = ld2(r, iv), Z;
      for (int e = 0; e < 2; ++e) {
        X.v[e] += alpha * P.v[e];
        const double ri = R.v[e] + (-alpha) * G.v[e];
        R.v[e] = ri; const double zi = ri * Mv.v[e]; Z.v[e] = zi; ztr += zi * ri;
        const double a = fabs(ri); mx = a > mx ? a : mx;
      }
      st2(x, iv, X); st2(r, iv, R);
      Pk[c] = P; Zk[c] = Z;
    }
  }
  ztr = bsum(ztr, red); mx = bmax(mx, red);
  if (threadIdx.x == 0) { pztr[blockIdx.x] = ztr; pmax[blockIdx.x] = mx; }
  if (!grid_barrier(bar, E, ctl)) return;
  double zs = 0, ms = 0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += BLK) { zs += __hip_atomic_load(&pztr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); const double v = __hip_atomic_load(&pmax[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ms = v > ms ? v : ms; }
  const double zt = bsum(zs, red), nr = bmax(ms, red);
  const bool conv = nr < tol;
  if (!conv) {
    const double beta = zt / zp;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int iv = gtid + c * gs;
      if (iv < nv) { d2 P = Pk[c]; for (int e = 0; e < 2; ++e) P.v[e] = Zk[c].v[e] + beta * P.v[e]; st2(p, iv, P); }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->ztr[par ^ 1] = zt; ctl->norm_r = nr; ctl->iters += 1; ctl->epoch = E; }
}
__global__ void k_barrier_only(Ctl *ctl, Bar *bar) {
  const unsigned E = ctl->epoch + 1;
  if (!grid_barrier(bar, E, ctl)) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->epoch = E;
}

template <typename T> T *devz(size_t n) { T *p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000000, reps = argc > 2 ? atoi(argv[2]) : 60;
  const size_t stream_doubles = argc > 3 ? (size_t)atof(argv[3]) : (n >= 1000000 ? 18750000 : (size_t)n * 19); // 150 MB at n = 1e6
  std::vector<double> h(n);
  auto fill = [&](double a, double b) { for (int i = 0; i < n; ++i) h[i] = a + b * ((i * 2654435761u) % 1000) / 1000.0; double *d; CK(hipMalloc(&d, (n + 8) * sizeof(double))); CK(hipMemcpy(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice)); return d; };
  double *x = fill(0, 1), *r = fill(-1, 2), *z = fill(0, 0), *p = fill(-0.5, 1), *Gp = fill(0.1, 1), *M = fill(0.5, 1);
  double *x2 = fill(0, 1), *r2 = fill(-1, 2), *p2 = fill(-0.5, 1);
  double *ppgp = devz<double>(4096), *pztr = devz<double>(4096), *pmax = devz<double>(4096), *sink = devz<double>(8);
  { std::vector<double> one(4096, 1.0 / 512); CK(hipMemcpy(ppgp, one.data(), 4096 * 8, hipMemcpyHostToDevice)); }
  double *A1 = devz<double>(stream_doubles + 8), *A2 = devz<double>(stream_doubles + 8);
  Ctl hc{}; hc.ztr[0] = hc.ztr[1] = 1e-3; hc.tol = 0; hc.done = 0;
  Ctl *ctl = devz<Ctl>(1), *ctl2 = devz<Ctl>(1); Bar *bar = devz<Bar>(1);
  CK(hipMemcpy(ctl, &hc, sizeof hc, hipMemcpyHostToDevice)); CK(hipMemcpy(ctl2, &hc, sizeof hc, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto body) { for (int w = 0; w < 5; ++w) body(w); CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) body(i); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / reps; };
  auto products = [&]() { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, st, A1, stream_doubles / 2, sink); hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, st, A2, stream_doubles / 2, sink); };
  printf("n=%d reps=%d stand-in products: 2 x %.0f MB\n", n, reps, stream_doubles * 8 / 1e6);
  const double t_prod = timeit([&](int) { products(); });
  printf("%-58s %8.2f us per iteration\n", "stand-in products alone", t_prod);
  for (int grid : {512, 256}) {
    const int gv = std::min(grid, (n / 2 + BLK - 1) / BLK);
    bar = devz<Bar>(1); // the arrival counters are monotonic for ONE grid size: fresh words (and epoch) per geometry
    ctl2 = devz<Ctl>(1); CK(hipMemcpy(ctl2, &hc, sizeof hc, hipMemcpyHostToDevice));
    const double t_two = timeit([&](int i) { products(); hipLaunchKernelGGL(k_update, dim3(gv), dim3(BLK), 0, st, x, r, z, p, Gp, M, n, ppgp, 512, pztr, pmax, ctl, i & 1); hipLaunchKernelGGL(k_direction, dim3(gv), dim3(BLK), 0, st, p, z, n, pztr, pmax, gv, ctl, i & 1); });
    printf("grid %4d  two kernels (update, direction)                 %8.2f us  (vector part %6.2f)\n", gv, t_two, t_two - t_prod);
    const int ch = (n / 2 + gv * BLK - 1) / (gv * BLK);
    double t_f = -1;
    auto fused = [&](int i) { products();
      switch (ch) {
      case 1: hipLaunchKernelGGL(k_updir<1>, dim3(gv), dim3(BLK), 0, st, x2, r2, p2, Gp, M, n, ppgp, 512, pztr, pmax, ctl2, bar, i & 1); break;
      case 2: hipLaunchKernelGGL(k_updir<2>, dim3(gv), dim3(BLK), 0, st, x2, r2, p2, Gp, M, n, ppgp, 512, pztr, pmax, ctl2, bar, i & 1); break;
      case 3: case 4: hipLaunchKernelGGL(k_updir<4>, dim3(gv), dim3(BLK), 0, st, x2, r2, p2, Gp, M, n, ppgp, 512, pztr, pmax, ctl2, bar, i & 1); break;
      default: hipLaunchKernelGGL(k_updir<8>, dim3(gv), dim3(BLK), 0, st, x2, r2, p2, Gp, M, n, ppgp, 512, pztr, pmax, ctl2, bar, i & 1); break;
      } };
    if (ch <= 8) t_f = timeit(fused);
    Ctl back; CK(hipMemcpy(&back, ctl2, sizeof back, hipMemcpyDeviceToHost));
    printf("grid %4d  fused behind one grid barrier (%d chunks/lane)    %8.2f us  (vector part %6.2f)  fault=%u epoch=%u\n", gv, ch, t_f, t_f - t_prod, back.fault, back.epoch);
    const double t_b = timeit([&](int) { hipLaunchKernelGGL(k_barrier_only, dim3(gv), dim3(BLK), 0, st, ctl2, bar); });
    printf("grid %4d  a kernel that is only the barrier                %8.2f us\n", gv, t_b);
  }
  // same data path check: both variants from identical state give identical p, x, r after one iteration
  {
    for (double *d : {x, x2}) { for (int i = 0; i < n; ++i) h[i] = (i % 97) * 0.01; CK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice)); }
    for (double *d : {r, r2}) { for (int i = 0; i < n; ++i) h[i] = ((i * 31) % 101) * 0.01 - 0.5; CK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice)); }
    for (double *d : {p, p2}) { for (int i = 0; i < n; ++i) h[i] = ((i * 17) % 89) * 0.01 - 0.4; CK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice)); }
    CK(hipMemcpy(ctl, &hc, sizeof hc, hipMemcpyHostToDevice));
    const int gv = std::min(512, (n / 2 + BLK - 1) / BLK), ch = (n / 2 + gv * BLK - 1) / (gv * BLK);
    bar = devz<Bar>(1); ctl2 = devz<Ctl>(1); CK(hipMemcpy(ctl2, &hc, sizeof hc, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_update, dim3(gv), dim3(BLK), 0, st, x, r, z, p, Gp, M, n, ppgp, 512, pztr, pmax, ctl, 0);
    hipLaunchKernelGGL(k_direction, dim3(gv), dim3(BLK), 0, st, p, z, n, pztr, pmax, gv, ctl, 0);
    if (ch <= 4) hipLaunchKernelGGL(k_updir<4>, dim3(gv), dim3(BLK), 0, st, x2, r2, p2, Gp, M, n, ppgp, 512, pztr, pmax, ctl2, bar, 0);
    else hipLaunchKernelGGL(k_updir<8>, dim3(gv), dim3(BLK), 0, st, x2, r2, p2, Gp, M, n, ppgp, 512, pztr, pmax, ctl2, bar, 0);
    CK(hipStreamSynchronize(st));
    std::vector<double> a(n), b(n); int bad = 0;
    for (auto pr : {std::pair<double *, double *>{x, x2}, {r, r2}, {p, p2}}) { CK(hipMemcpy(a.data(), pr.first, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), pr.second, n * 8, hipMemcpyDeviceToHost)); for (int i = 0; i < n; ++i) bad += a[i] != b[i]; }
    printf("bitwise comparison of x, r, p after one iteration (two kernels vs fused): %d differing entries\n", bad);
  }
  return 0;
}
