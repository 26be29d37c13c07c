// spmv_sliced.h -- column-sliced, time-aligned CSR product for gather vectors that do
// not fit a 4 MB XCD L2 (same results and epilogues as csr_stream_kernel in spmv.h).
//
// Why: on the headline config the product is gather-bound, not stream-bound.  Measured
// on MI355X (lab/spmv_lab.hip, 1e7 random 8-byte gathers): 37 us when the table is
// L2-resident (<= 4 MB), 80 us at 8 MB, 111 us at 16 MB, against 17 us to stream the
// 120 MB of val/idx.  The vectors gathered here are 8 MB (p) and 16 MB (R_y^-1 A p).
//
// How: a workgroup owns a "super-block" of consecutive rows holding ~8192 nonzeros and
// keeps one accumulator per row in LDS.  Inside the super-block the entries are stored
// sorted by (column slice, row), a slice being 2^16 consecutive columns (512 KB of fp64
// x).  Every workgroup walks the slices in the same order starting at the same time, so
// at any moment the workgroups sharing an XCD gather from the same one or two slices of
// x -- L2-resident -- without any barrier between workgroups.  Per entry the format
// stores val (8 B) and one packed word (4 B): column-within-slice | local row << 16, i.e.
// the same 12 B/nnz as CSR and no row-pointer array.  A chunk of products goes to LDS;
// the first lane of each (row, slice) run adds the run to its row accumulator (runs are
// contiguous because of the sort, so no two lanes touch the same accumulator:
// deterministic, no atomics).  Summation order inside a row is slice-major instead of
// the reference's index order (rounding-level difference only).
// Measured: 62-65 us per product in either orientation vs 111/150 us for csr_stream.
#pragma once
#include "spmv.h"
#include <algorithm>

namespace scsamd {

// columns per slice: 512 KB of x either way (2^16 doubles / 2^17 floats)
constexpr int SL_SLICE_BITS = sizeof(real) == 8 ? 16 : 17;
constexpr int SL_NNZ_SB = 8192;    // nonzeros per super-block (minimum target)
constexpr int SL_ROWS_MAX = 32768 / sizeof(real); // rows per super-block: 32 KB of LDS accumulators
constexpr int SL_TARGET_WGS = 1400; // keep every workgroup resident (time alignment needs one wave of WGs)
constexpr int SL_CHUNK = 512;      // products staged per step
constexpr int SL_MAX_GRID = 16384;

struct SlicedView {
  int rows, nsb, S, bits; // bits: log2(columns per slice)
  const int *sbrow;       // nsb + 1 : first row of each super-block
  const int *segoff;      // nsb * (S + 1) : start of each (super-block, slice) run
  const unsigned *sidx;   // nnz : (col & 0xffff) | local_row << 16
  const real *sval;       // nnz
};

#ifdef __HIPCC__
template <int EPI>
__global__ __launch_bounds__(SCSAMD_BLOCK) void csr_sliced_kernel(SlicedView A, const real *__restrict__ x, real *y,
                                                                  EpiArgs e, const int *skip, int accrows) {
  if (skip && *skip) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char sl_smem[];
  real *acc = reinterpret_cast<real *>(sl_smem);
  real *sp = acc + accrows;
  unsigned short *sr = reinterpret_cast<unsigned short *>(sp + SL_CHUNK);
  real *red = reinterpret_cast<real *>(sr + SL_CHUNK); // 8-byte aligned: SL_CHUNK * 2 B is a multiple of 8
  constexpr int U = SL_CHUNK / SCSAMD_BLOCK;
  const int tid = threadIdx.x;
  const int bits = A.bits;
  const unsigned mask = (1u << bits) - 1;
  real dot = 0;
  for (int b = blockIdx.x; b < A.nsb; b += gridDim.x) {
    const int r0 = A.sbrow[b], nr = A.sbrow[b + 1] - r0;
    for (int r = tid; r < nr; r += SCSAMD_BLOCK) acc[r] = 0;
    __syncthreads();
    const int *so = A.segoff + (size_t)b * (A.S + 1);
    for (int s = 0; s < A.S; ++s) {
      const int e0 = so[s], e1 = so[s + 1];
      const real *xs = x + ((size_t)s << bits);
      for (int base = e0; base < e1; base += SL_CHUNK) {
        const int cnt = min(SL_CHUNK, e1 - base);
        unsigned w[U];
        real v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int k = tid + j * SCSAMD_BLOCK;
          const bool ok = k < cnt;
          w[j] = ok ? A.sidx[base + k] : 0u;
          v[j] = ok ? A.sval[base + k] : (real)0;
        }
        real xx[U];
#pragma unroll
        for (int j = 0; j < U; ++j) xx[j] = xs[w[j] & mask];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int k = tid + j * SCSAMD_BLOCK;
          if (k < cnt) {
            sp[k] = v[j] * xx[j];
            sr[k] = (unsigned short)(w[j] >> bits);
          }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int k = tid + j * SCSAMD_BLOCK;
          if (k < cnt) {
            const unsigned short lr = sr[k];
            if (k == 0 || sr[k - 1] != lr) { // first lane of a (row, slice) run owns it
              real sum = sp[k];
              int kk = k + 1;
              while (kk < cnt && sr[kk] == lr) sum += sp[kk++];
              acc[lr] += sum;
            }
          }
        }
        __syncthreads();
      }
    }
    for (int r = tid; r < nr; r += SCSAMD_BLOCK) {
      const real a = epi_init<EPI>(e, y, r0 + r) + acc[r];
      epi_apply<EPI>(e, y, r0 + r, a, dot);
    }
    __syncthreads();
  }
  if (EPI == EPI_GP && e.partial) {
    dot = block_sum(dot, red);
    if (tid == 0) e.partial[blockIdx.x] = dot;
  }
}
#endif // __HIPCC__

struct SlicedDev {
  bool built = false;
  int rows = 0, cols = 0, nsb = 0, S = 0, accrows = 0, bits = SL_SLICE_BITS;
  DevBuf<int> sbrow, segoff;
  DevBuf<unsigned> sidx;
  DevBuf<real> sval;
  SlicedView view() const { return SlicedView{rows, nsb, S, bits, sbrow.p, segoff.p, sidx.p, sval.p}; }
  int grid() const { return std::max(1, std::min(nsb, SL_MAX_GRID)); }
  size_t lds_bytes() const {
    return (size_t)accrows * sizeof(real) + SL_CHUNK * (sizeof(real) + sizeof(unsigned short)) + 8 * sizeof(real);
  }
  // worth it only when the gathered vector overflows an XCD's L2 and rows are short
  static bool wanted(int cols, const int *hptr, int rows) {
    if (const char *e = getenv("SCS_AMD_SLICED")) return atoi(e) != 0; // tests force either path
    // pays off only when x clearly overflows the 4 MB L2 of an XCD (measured: an 8 MB table halves
    // the gather rate) and there are enough super-blocks to fill 256 CUs
    if ((size_t)cols * sizeof(real) <= (size_t)6 << 20) return false;
    if ((long long)hptr[rows] < 4LL * SL_NNZ_SB * 128) return false;
    long long mx = 0;
    for (int r = 0; r < rows; ++r) mx = std::max<long long>(mx, hptr[r + 1] - hptr[r]);
    return mx <= 4 * SL_NNZ_SB;
  }
  void build(int rows_, int cols_, const int *hptr, const int *hidx, const real *hval, hipStream_t st) {
    rows = rows_;
    cols = cols_;
    bits = SL_SLICE_BITS;
    if (const char *e = getenv("SCS_AMD_SLICE_BITS")) bits = std::max(10, std::min(19, atoi(e)));
    S = std::max(1, (cols + (1 << bits) - 1) >> bits);
    // Super-block boundaries for a given nonzero budget (greedy over consecutive rows).
    auto partition = [&](long long budget, std::vector<int> &out) {
      out.clear();
      out.push_back(0);
      int r = 0;
      while (r < rows) {
        const int s0 = r;
        long long acc = 0;
        while (r < rows && r - s0 < SL_ROWS_MAX) {
          const long long rn = hptr[r + 1] - hptr[r];
          if (acc + rn > budget && r > s0) break;
          acc += rn;
          ++r;
        }
        out.push_back(r);
      }
    };
    // All workgroups are resident at once and the per-CU L1 fill rate is the limit, so a launch
    // whose workgroup count is not a multiple of the CU count finishes with its fullest CUs
    // (measured: 1222 workgroups on 256 CUs 84.8 us, 1024 or 1280 workgroups 82.7 us).  Pick the
    // multiple of the CU count closest to the ~8192-nonzero budget (at most SL_TARGET_WGS).
    std::vector<int> sb;
    const long long nnz_all = hptr[rows];
    long long nnz_sb = std::max<long long>(SL_NNZ_SB, (nnz_all + SL_TARGET_WGS - 1) / SL_TARGET_WGS);
    if (const char *e = getenv("SCS_AMD_SL_NNZ_SB")) {
      nnz_sb = std::max(1024, atoi(e)); // experiments
      partition(nnz_sb, sb);
    } else {
      int cus = 256, dev = 0;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      cus = std::max(1, cus);
      long long k = (nnz_all / SL_NNZ_SB + cus / 2) / cus; // workgroups per CU, rounded
      k = std::max<long long>(1, std::min<long long>(k, SL_TARGET_WGS / cus));
      const long long want = k * cus;
      nnz_sb = std::max<long long>(1024, (nnz_all + want - 1) / want);
      partition(nnz_sb, sb);
      for (int tries = 0; (long long)sb.size() - 1 > want && tries < 40; ++tries) { // greedy packing overshoots a little
        nnz_sb += std::max<long long>(1, nnz_sb / 200);
        partition(nnz_sb, sb);
      }
    }
    nsb = (int)sb.size() - 1;
    accrows = 2;
    for (int b = 0; b < nsb; ++b) accrows = std::max(accrows, sb[b + 1] - sb[b]);
    accrows = (accrows + 1) & ~1;
    const size_t nnz = (size_t)hptr[rows];
    std::vector<unsigned> hi(nnz ? nnz : 1);
    std::vector<real> hv(nnz ? nnz : 1);
    std::vector<int> so((size_t)nsb * (S + 1));
    std::vector<int> cnt(S + 1), nx(S);
    for (int b = 0; b < nsb; ++b) {
      const int k0 = hptr[sb[b]], k1 = hptr[sb[b + 1]];
      std::fill(cnt.begin(), cnt.end(), 0);
      for (int k = k0; k < k1; ++k) cnt[(hidx[k] >> bits) + 1]++;
      for (int s = 0; s < S; ++s) cnt[s + 1] += cnt[s];
      for (int s = 0; s <= S; ++s) so[(size_t)b * (S + 1) + s] = k0 + cnt[s];
      for (int s = 0; s < S; ++s) nx[s] = cnt[s];
      for (int rr = sb[b]; rr < sb[b + 1]; ++rr)
        for (int k = hptr[rr]; k < hptr[rr + 1]; ++k) {
          const int s = hidx[k] >> bits;
          const size_t q = (size_t)k0 + nx[s]++;
          hi[q] = (unsigned)(hidx[k] & ((1 << bits) - 1)) | ((unsigned)(rr - sb[b]) << bits);
          hv[q] = hval[k];
        }
    }
    sbrow.alloc(sb.size());
    segoff.alloc(so.size());
    sidx.alloc(nnz ? nnz : 1);
    sval.alloc(nnz ? nnz : 1);
    sbrow.upload(sb.data(), sb.size(), st);
    segoff.upload(so.data(), so.size(), st);
    if (nnz) {
      sidx.upload(hi.data(), nnz, st);
      sval.upload(hv.data(), nnz, st);
    }
    HIP_CHECK(hipStreamSynchronize(st));
    built = true;
  }
};

} // namespace scsamd
