// spmv_wave_build.h -- the wave-owned-rows layout of spmv_wave.h built ON THE DEVICE (round 5): the values and
// packed words of every unit in column-bucket order (+ the quarter-window chunk order of the lockstep kernel) and the count of
// distinct 128-byte lines a unit gathers from, straight from the CSR copy that is already in HBM.  Bit-identical to
// WaveRowsDev::fill_host (the host builder of rounds 2-4, which stays as the general path and as this builder's oracle:
// SCS_AMD_WR_BUILD = dev | host | verify; `verify` builds both and compares every byte -- tests/test_wave_build_gpu.py).
// Replaces the host work the reference does in linsys/cpu/indirect/private.c:7-46 (its transpose) / the layout conversions of
// linsys/gpu/indirect/private.c:328-344 with device work, as the reference's own GPU backend does.
//
// One workgroup per unit (a unit = the <= 1024 consecutive rows one wave owns; its entries fit LDS):
//   1. key = (column bucket << 13) | t, t = the entry's position in the unit's CSR order; an in-LDS bitonic sort of these unique
//      keys IS the host's stable counting sort by bucket (ties in row-major order);
//   2. lockstep layout only: inside every 256-entry chunk of that order, key = (column << 21) | (position in chunk << 13) | t,
//      chunk-local bitonic sort = the host's stable sort by column; rank r of a chunk then goes to position 4 (r % 64) + r / 64
//      (short last chunk: ranks dealt to the valid positions in the same i-major order as the host);
//   3. distinct lines of the gathered vector: a bitmap in LDS, popcount, one integer atomic per unit (order independent).
#pragma once
#include "spmv_wave.h"

namespace scsamd {

constexpr int WR_DEV_UNIT_MAX = 8192; // entries of a unit the device builder sorts in LDS (13 bits of t); longer units: host builder
constexpr int WB_THREADS = 512;
constexpr size_t WB_LDS_KEYS = (size_t)WR_DEV_UNIT_MAX * 8;  // 64-bit keys of phase 2 (phase 1's 32-bit keys use the first half)
constexpr size_t WB_LDS_ROWL = (size_t)WR_DEV_UNIT_MAX * 2;  // local row of every entry
constexpr size_t WB_LDS_MAX = 150 * 1024;

#ifdef __HIPCC__
// ascending bitonic sort of d[0, N), N a power of two; CHUNK > 0: every aligned block of CHUNK elements sorted on its own
template <typename K, int CHUNK>
__device__ __forceinline__ void wb_bitonic(K *d, int N, int tid) {
  const int kmax = CHUNK > 0 ? (CHUNK < N ? CHUNK : N) : N;
  for (int k = 2; k <= kmax; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < N; i += WB_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool asc = k == kmax ? true : (i & k) == 0; // the last merge of every block runs upwards
          const K a = d[i], b = d[ixj];
          if ((a > b) == asc) {
            d[i] = b;
            d[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

// position of rank r among the `len` valid entries of a chunk: ranks are dealt to the positions 4 l + i in the order i = 0 (l = 0, 1,
// ...), i = 1, ... skipping positions >= len (WaveRowsDev::fill_host)
__device__ __forceinline__ int wb_deal(int r, int len) {
  int cum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = len > i ? (len - i + 3) >> 2 : 0;
    if (r < cum + c) return 4 * (r - cum) + i;
    cum += c;
  }
  return 0; // not reached for r < len
}

template <bool SUBWIN>
__global__ __launch_bounds__(WB_THREADS) void k_wave_layout(const eoff *__restrict__ ptr, const int *__restrict__ idx,
                                                            const real *__restrict__ val, const int *__restrict__ urow,
                                                            const eoff *__restrict__ useg, unsigned *wrd, real *vout, int cbits,
                                                            int bshift, int lshift, int bm_words, unsigned long long *distinct,
                                                            unsigned short *rowl_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wb_smem[];
  unsigned long long *key64 = reinterpret_cast<unsigned long long *>(wb_smem);
  unsigned *key32 = reinterpret_cast<unsigned *>(wb_smem);
  unsigned short *rowl = reinterpret_cast<unsigned short *>(wb_smem + WB_LDS_KEYS);
  unsigned *bm = reinterpret_cast<unsigned *>(wb_smem + WB_LDS_KEYS + WB_LDS_ROWL);
  __shared__ unsigned red[WB_THREADS / 64];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int r0 = urow[u], r1 = urow[u + 1];
  const eoff k0 = ptr[r0], k1 = ptr[r1], base = useg[2 * u];
  const int len = (int)(k1 - k0); // <= WR_DEV_UNIT_MAX (checked on the host)
  if (len <= 0) return; // uniform
  int P2 = 256;
  while (P2 < len) P2 <<= 1;
  for (int w = tid; w < bm_words; w += WB_THREADS) bm[w] = 0;
  for (int rr = r0 + tid; rr < r1; rr += WB_THREADS)
    for (eoff k = ptr[rr]; k < ptr[rr + 1]; ++k) rowl[k - k0] = (unsigned short)(rr - r0);
  __syncthreads();
  for (int t = tid; t < P2; t += WB_THREADS) {
    if (t < len) {
      const unsigned col = (unsigned)idx[k0 + t];
      key32[t] = ((col >> bshift) << 13) | (unsigned)t;
      if (bm_words > 0) { // (beyond WaveRowsDev::lines_counted the lines are not counted: one per entry is reported)
        const unsigned line = col >> lshift;
        atomicOr(&bm[line >> 5], 1u << (line & 31));
      }
    } else {
      key32[t] = 0xFFFFFFFFu;
    }
  }
  __syncthreads();
  wb_bitonic<unsigned, 0>(key32, P2, tid);
  { // distinct lines this unit gathers from
    unsigned c = 0;
    for (int w = tid; w < bm_words; w += WB_THREADS) c += __popc(bm[w]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
      unsigned tot = 0;
      for (int w = 0; w < WB_THREADS / 64; ++w) tot += red[w];
      atomicAdd(distinct, bm_words > 0 ? (unsigned long long)tot : (unsigned long long)len);
    }
  }
  if (!SUBWIN) {
    for (int r = tid; r < len; r += WB_THREADS) {
      const int t = (int)(key32[r] & 8191u);
      if (rowl_out) { // wide layout: the column alone, the local row in its own array
        wrd[base + r] = (unsigned)idx[k0 + t];
        rowl_out[base + r] = rowl[t];
      } else {
        wrd[base + r] = (unsigned)idx[k0 + t] | ((unsigned)rowl[t] << cbits);
      }
      vout[base + r] = val[k0 + t];
    }
    return;
  }
  // phase 2: inside every 256-entry chunk by (column, position in the chunk)
  constexpr int PER = WR_DEV_UNIT_MAX / WB_THREADS; // 16
  unsigned long long kk[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int r = tid + j * WB_THREADS;
    kk[j] = ~0ull;
    if (r < len) {
      const unsigned t = key32[r] & 8191u;
      kk[j] = ((unsigned long long)(unsigned)idx[k0 + (int)t] << 21) | ((unsigned long long)(r & 255) << 13) | t;
    }
  }
  __syncthreads(); // every 32-bit key has been read before the 64-bit keys overwrite them
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int r = tid + j * WB_THREADS;
    if (r < P2) key64[r] = kk[j];
  }
  __syncthreads();
  wb_bitonic<unsigned long long, 256>(key64, P2, tid);
  for (int r = tid; r < len; r += WB_THREADS) {
    const int c0 = r & ~255, clen = len - c0 < 256 ? len - c0 : 256;
    const unsigned long long key = key64[r];
    const int t = (int)(key & 8191u);
    const eoff o = base + c0 + wb_deal(r & 255, clen);
    wrd[o] = (unsigned)(key >> 21) | ((unsigned)rowl[t] << cbits);
    vout[o] = val[k0 + t];
  }
}

// fills w.wrd / w.val (allocated, zeroed) from the device CSR arrays; false = this matrix needs the host builder
inline bool wave_fill_dev(WaveRowsDev &w, const eoff *d_ptr, const int *d_idx, const real *d_val, hipStream_t st, long long &distinct) {
  const int lshift = sizeof(real) == 8 ? 4 : 5; // 128-byte line = 16 fp64 / 32 fp32 entries
  const long long lines = ((long long)w.cols >> lshift) + 1;
  const int bm_words = WaveRowsDev::lines_counted(w.cols) ? (int)((lines + 31) / 32) : 0;
  if (w.wide && w.sub_window_order) return false; // (never planned: the wide layout runs the plain schedule)
  const size_t lds = WB_LDS_KEYS + WB_LDS_ROWL + (size_t)bm_words * 4;
  if (w.max_unit_entries() > WR_DEV_UNIT_MAX || lds > WB_LDS_MAX || w.nunit < 1) return false;
  // the limit that counts is the device's opt-in LDS per workgroup (queried, not assumed), and the kernel's static LDS
  // (its reduction array) counts against it too; a refused opt-in sends the matrix to the host builder instead of failing scs_init
  {
    int dev = 0, optin = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return false;
    hipFuncAttributes fa;
    const void *kp = w.sub_window_order ? reinterpret_cast<const void *>(k_wave_layout<true>) : reinterpret_cast<const void *>(k_wave_layout<false>);
    if (hipFuncGetAttributes(&fa, kp) != hipSuccess) return false;
    if (lds + fa.sharedSizeBytes > (size_t)optin) return false;
    if (hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
  }
  DevBuf<unsigned long long> cnt(1);
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(w.nunit), dim3(WB_THREADS), lds, st, d_ptr, d_idx, d_val, (const int *)w.urow.p, (const eoff *)w.useg.p,
                       w.wrd.p, w.val.p, w.cbits, w.bshift, lshift, bm_words, cnt.p, w.wide ? w.rowl.p : (unsigned short *)nullptr);
  };
  if (w.sub_window_order) launch(k_wave_layout<true>);
  else launch(k_wave_layout<false>);
  HIP_CHECK(hipGetLastError());
  unsigned long long h = 0;
  HIP_CHECK(hipMemcpyAsync(&h, cnt.p, sizeof h, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  distinct = (long long)h;
  return true;
}

// the layout of `mat` (already uploaded as CSR): on the device when it can be, on the host otherwise; SCS_AMD_WR_BUILD = host | dev |
// verify (both, compared byte for byte -- throws on any difference)
inline void wave_build(WaveRowsDev &w, int rows, int cols, const eoff *hptr, const int *hidx, const real *hval, const CsrDev &mat,
                       hipStream_t st) {
  int mode = 1; // dev
  if (const char *e = opt_get("wr_build")) mode = !strcmp(e, "host") ? 0 : (!strcmp(e, "verify") ? 2 : 1);
  w.plan(rows, cols, hptr);
  w.alloc_and_upload_plan(st);
  long long distinct = 0;
  bool on_dev = false;
  if (mode != 0) on_dev = wave_fill_dev(w, mat.ptr.p, mat.idx.p, mat.val.p, st, distinct);
  w.built_on_device = on_dev;
  if (!on_dev || mode == 2) {
    std::vector<unsigned> hw;
    std::vector<real> hv;
    std::vector<int> fetched_idx;
    std::vector<real> fetched_val;
    if (!hidx || !hval) { // the CSR arrays were adopted from HBM: fetch what the host builder reads
      const size_t nz = (size_t)hptr[rows];
      if (!hidx) {
        fetched_idx.resize(nz);
        mat.idx.download(fetched_idx.data(), nz, st);
        hidx = fetched_idx.data();
      }
      if (!hval) {
        fetched_val.resize(nz);
        mat.val.download(fetched_val.data(), nz, st);
        hval = fetched_val.data();
      }
      HIP_CHECK(hipStreamSynchronize(st));
    }
    long long dh = 0;
    std::vector<unsigned short> hr;
    w.fill_host(hptr, hidx, hval, hw, hv, dh, &hr);
    if (on_dev) { // verify
      std::vector<unsigned> gw(w.cap);
      std::vector<real> gv(w.cap);
      std::vector<unsigned short> gr(w.wide ? w.cap : 0);
      w.wrd.download(gw.data(), w.cap, st);
      w.val.download(gv.data(), w.cap, st);
      if (w.wide) w.rowl.download(gr.data(), w.cap, st);
      HIP_CHECK(hipStreamSynchronize(st));
      if (dh != distinct || memcmp(gw.data(), hw.data(), w.cap * sizeof(unsigned)) != 0 || memcmp(gv.data(), hv.data(), w.cap * sizeof(real)) != 0 ||
          (w.wide && memcmp(gr.data(), hr.data(), w.cap * sizeof(unsigned short)) != 0))
        throw HipError("scs_amd: device-built wave-rows layout differs from the host builder's (SCS_AMD_WR_BUILD=verify)");
    } else {
      w.wrd.upload(hw.data(), w.cap, st);
      w.val.upload(hv.data(), w.cap, st);
      if (w.wide) w.rowl.upload(hr.data(), w.cap, st);
      HIP_CHECK(hipStreamSynchronize(st));
      distinct = dh;
    }
  }
  w.finish(distinct);
}
#endif // __HIPCC__

} // namespace scsamd
