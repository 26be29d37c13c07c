// common.h -- shared host/device helpers for the MI355X (gfx950) SCS hot path.
//
// Conventions used by every kernel in this directory:
//  * wavefront = 64 lanes, workgroup = 256 threads (4 waves, one per SIMD);
//  * every floating-point reduction is two-level and deterministic: producer
//    workgroups write one partial each (fixed grid, fixed accumulation order),
//    and every consumer workgroup re-reduces the whole partial array in the
//    same fixed order (a few KB out of L2) -- no atomics, no host round trip;
//  * kernels that belong to a device-controlled loop (PCG, box Newton) take a
//    control block and return immediately once its `done` word is set, so the
//    host can enqueue iterations speculatively and poll one word per batch.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <stdexcept>
#include <atomic>

#include "../../include/scs_amd.h"
#include "options.h"

namespace scsamd {

// Position of an entry in the value / index arrays of a sparse matrix (row pointers, unit entry ranges, CSC positions).  32-bit
// like every other index -- except in the DLONG build (64-bit scs_int at the ABI, include/scs_types.h:13-20), where a matrix may
// hold 2^31 nonzeros or more (24 GB of CSR per orientation against 288 GB of HBM).  Column / row INDICES stay 32-bit in both
// builds (m, n < 2^31).  Round 5.
#ifdef DLONG
typedef long long eoff;
#else
typedef int eoff;
#endif

#define SCSAMD_WAVE 64
#define SCSAMD_BLOCK 256

// ---- error handling: HIP failures become C++ exceptions that the extern "C"
// layer maps onto the reference's conventions (NULL / non-zero / <0).
struct HipError : std::runtime_error {
  explicit HipError(const std::string &m) : std::runtime_error(m) {}
};
// Fault injection (tests of the failure convention): every HIP runtime call of this library goes through
// HIP_CHECK, so ONE countdown covers allocations, copies, synchronisations and the post-launch error polls.  Armed by
// the explicit test entry point scs_amd_test_fail_at(k) ONLY (round 5, ADVICE r4: no environment variable arms it -- a stray
// variable must not make a production HIP call report OOM): the k-th checked call from then on is reported as
// hipErrorOutOfMemory although it succeeded; the countdown then disarms itself.  One relaxed atomic load per checked call.
inline std::atomic<long long> &fail_countdown() {
  static std::atomic<long long> c{0LL};
  return c;
}
inline void hip_check(hipError_t e, const char *what, const char *file, int line) {
  std::atomic<long long> &cd = fail_countdown();
  if (cd.load(std::memory_order_relaxed) > 0 && cd.fetch_sub(1, std::memory_order_relaxed) == 1 && e == hipSuccess)
    e = hipErrorOutOfMemory; // injected
  if (e != hipSuccess) {
    char buf[512];
    snprintf(buf, sizeof buf, "scs_amd: HIP error %d (%s) at %s:%d in %s", (int)e,
             hipGetErrorString(e), file, line, what);
    throw HipError(buf);
  }
}
#define HIP_CHECK(x) ::scsamd::hip_check((x), #x, __FILE__, __LINE__)

// one zero-fill stream per host thread, owned by a thread_local holder: destroyed when the thread
// exits and when the thread switches device
struct FillStream {
  hipStream_t st = nullptr;
  int dev = -1;
  ~FillStream() {
    if (st) (void)hipStreamDestroy(st);
  }
};
inline hipStream_t fill_stream() {
  static thread_local FillStream fs;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  if (!fs.st || fs.dev != dev) {
    if (fs.st) {
      (void)hipStreamDestroy(fs.st); // queued fills complete first: hipStreamDestroy is asynchronous-safe
      fs.st = nullptr;
    }
    HIP_CHECK(hipStreamCreateWithFlags(&fs.st, hipStreamNonBlocking));
    fs.dev = dev;
  }
  return fs.st;
}

// ---- device buffer (RAII) -------------------------------------------------
template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  // ownership can be handed on (setup hands matrices that are already in HBM to the solver instead of re-uploading them)
  void take(DevBuf &o) {
    release();
    p = o.p;
    n = o.n;
    o.p = nullptr;
    o.n = 0;
  }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    // always allocate a little slack so 16-byte vector loads that start inside
    // the array may run a few elements past its logical end
    HIP_CHECK(hipMalloc((void **)&p, (count + 8) * sizeof(T)));
    // Zero-fill on a private non-blocking stream and wait for it: the fill must have happened
    // before any work stream touches the buffer (they have no implicit ordering with it), and it
    // must NOT go through the legacy default stream -- any legacy-stream operation fails with
    // hipErrorStreamCaptureImplicit while another host thread is capturing a HIP graph (the PCG
    // loop of a concurrent solve does exactly that).
    hipStream_t fs = fill_stream();
    HIP_CHECK(hipMemsetAsync(p, 0, (count + 8) * sizeof(T), fs));
    HIP_CHECK(hipStreamSynchronize(fs));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void upload(const T *h, size_t count, hipStream_t s) {
    HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T *h, size_t count, hipStream_t s) const {
    HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
  }
};

// ---- pinned host scalar block ----------------------------------------------
template <typename T> struct PinnedBuf {
  T *p = nullptr;
  size_t n = 0;
  PinnedBuf() = default;
  explicit PinnedBuf(size_t count) { alloc(count); }
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void alloc(size_t count) {
    if (p) (void)hipHostFree(p);
    n = count;
    HIP_CHECK(hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault));
    memset(p, 0, count * sizeof(T));
  }
};

// ---- HIP-event stopwatch pool (sampled kernel timing on OUR stream) --------
struct EventTimer {
  std::vector<hipEvent_t> a, b;
  size_t used = 0;
  double total_ms = 0.0;
  long long samples = 0;
  ~EventTimer() {
    for (auto e : a) (void)hipEventDestroy(e);
    for (auto e : b) (void)hipEventDestroy(e);
  }
  // returns slot or -1
  int start(hipStream_t s) {
    if (used == a.size()) {
      if (a.size() >= 512) return -1;
      hipEvent_t e0, e1;
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      a.push_back(e0);
      b.push_back(e1);
    }
    HIP_CHECK(hipEventRecord(a[used], s));
    return (int)used++;
  }
  void stop(int slot, hipStream_t s) {
    if (slot >= 0) HIP_CHECK(hipEventRecord(b[slot], s));
  }
  // call only when the stream is known to be idle
  void harvest() {
    for (size_t i = 0; i < used; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, a[i], b[i]) == hipSuccess) {
        total_ms += ms;
        samples++;
      }
    }
    used = 0;
  }
};

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// ---- device-side reductions (deterministic, fixed order) -------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v; // valid in lane 0
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_down(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}
// Every thread of the workgroup receives the same sum.  `sh` needs
// blockDim.x/64 entries; may be reused right after return.
template <typename T> __device__ __forceinline__ T block_sum(T v, T *sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  T s = sh[0];
  for (int i = 1; i < nw; ++i) s += sh[i];
  return s;
}
template <typename T> __device__ __forceinline__ T block_max(T v, T *sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  T s = sh[0];
  for (int i = 1; i < nw; ++i) s = sh[i] > s ? sh[i] : s;
  return s;
}
// consumer side of the two-level reduction: every workgroup re-reduces the
// producer's partial array (count entries) in a fixed order.
template <typename T>
__device__ __forceinline__ T reduce_partials_sum(const T *part, int count, T *sh) {
  T s = 0;
  for (int i = threadIdx.x; i < count; i += blockDim.x) s += part[i];
  return block_sum(s, sh);
}
template <typename T>
__device__ __forceinline__ T reduce_partials_max(const T *part, int count, T *sh) {
  T s = 0;
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    T v = part[i];
    s = v > s ? v : s;
  }
  return block_max(s, sh);
}
template <typename T> __device__ __forceinline__ T absval(T x) { return x < 0 ? -x : x; }
// 16-byte vector access for the streaming elementwise kernels (dwordx4 per lane)
constexpr int RVW = 16 / sizeof(scs_float);
struct alignas(16) rvec {
  scs_float v[RVW];
};
__device__ __forceinline__ rvec ldv(const scs_float *p, size_t iv) { return reinterpret_cast<const rvec *>(p)[iv]; }
__device__ __forceinline__ void stv(scs_float *p, size_t iv, const rvec &x) { reinterpret_cast<rvec *>(p)[iv] = x; }
// the same with the non-temporal cache policy (streams that nothing re-reads soon: they should not push the matrix streams and
// the gathered vectors out of L2 / the Infinity Cache)
typedef scs_float rvec_nt __attribute__((ext_vector_type(RVW)));
__device__ __forceinline__ rvec ldv_nt(const scs_float *p, size_t iv) {
  const rvec_nt v = __builtin_nontemporal_load(reinterpret_cast<const rvec_nt *>(p) + iv);
  rvec r;
#pragma unroll
  for (int e = 0; e < RVW; ++e) r.v[e] = v[e];
  return r;
}
__device__ __forceinline__ void stv_nt(scs_float *p, size_t iv, const rvec &x) {
  rvec_nt v;
#pragma unroll
  for (int e = 0; e < RVW; ++e) v[e] = x.v[e];
  __builtin_nontemporal_store(v, reinterpret_cast<rvec_nt *>(p) + iv);
}
#endif // __HIPCC__

} // namespace scsamd
