// spmv_cusort.h -- "CU-sorted" CSR product: a round-4 experiment that was MEASURED AND NOT KEPT (kept here, outside the build, as the record).
// The wave-owned-rows product of scs_amd/csrc/spmv_wave.h with the accumulators of a whole CU in ONE LDS array and the CU's entries in ONE
// column order.  It was wired in behind SCS_AMD_WR_CUSORT=1 (WaveRowsDev::build calling CuSortDev::build_host, a launch_cusort<E> next to
// launch_lockstep<E> in linsys.hip), passed tests/test_linsys_gpu.py and the host emulation below (bit-identical to the dealing order, error 0
// against a dense product at n = 1e6), and measured on the headline matrix (n=1e6, m=2e6, nnz=1e7, fp64; gpurun call M, scripts/gpu_r4_m.sh):
//     lockstep kernel (the default)  61.9 / 62.1 us per product, 156.6 / 157.1 us per CG iteration
//     this kernel                    66.4 / 66.5 us per product, 157.3 / 157.7 us per CG iteration
// 4.4 % padding slots, 12.8 % (A') / 6.6 % (A) of the entries deferred by a round, two workgroup barriers per round instead of one: the
// ~20 % fewer distinct lines per gather instruction do not pay for them -- consistent with profiles/r4_spmv_lockstep.md (the L1->L2 request
// count is not what bounds the product; the alignment of the 16 waves' gather phases is).
//
// Why: a CU's ~39 K gathers per product touch only ~29 K (A) / ~33 K (A') distinct 128-byte lines of the gathered vector, but with one
// column order PER WAVE a 64-lane gather instruction covers a window of ~1600 lines with 64 gathers -- every lane its own line.  With one
// column order per CU, the 1024 gathers the CU's 16 waves issue together are 1024 CONSECUTIVE entries of that order: wave w's instruction
// covers a sixteenth of the window with 64 gathers, so lanes of one instruction share lines and the memory pipeline sees ~20 % fewer
// requests (the resource everything queues for).
//
// How the sums stay deterministic with accumulators shared by 16 waves: LDS adds of ONE wave execute in program order; adds of different
// waves do not.  The host therefore deals the entries so that within one round (= one gather instruction of all 16 waves = 1024 slots)
// all entries of a row sit in the SAME wave -- an entry whose row already sits in another wave of the round is deferred to the next round
// (6 % of the entries on the headline matrix) -- and the kernel separates the rounds' adds by workgroup barriers.  Per row the order is
// then: round by round, inside a round the lane order of one instruction -- fixed.
//
// Layout per CU unit (a range of rows balanced by nonzeros, <= 16384 rows): chunks of 4096 slots = 4 rounds; slot q of round i of a chunk
// sits at chunk * 4096 + 4 q + i (a lane's 4 rounds are contiguous: 16-byte loads); word = (column - chunk base) | local row << cbits;
// padding slots carry value 0 and add +0.0 to row 0.
#pragma once
// (included by spmv_wave.h behind its lds_add)
#include <algorithm>

namespace scsamd {

constexpr int CS_WAVES = 16;
constexpr int CS_BLOCK = CS_WAVES * 64;
constexpr int CS_ROUND = CS_BLOCK;      // slots per round
constexpr int CS_CHUNK = 4 * CS_ROUND;  // slots per chunk
constexpr int CS_ROWS_MAX = 16384;      // 128 KB of fp64 accumulators

struct CuSortView {
  int rows, nunit, cbits;
  const int *urow;      // nunit + 1: first row of each unit
  const int *uchunk;    // nunit + 1: first chunk of each unit
  const int *cbase;     // per chunk: column base
  const unsigned *wrd;  // per slot
  const real *val;      // per slot
};

#if defined(__HIPCC__) && !defined(SFLOAT)
template <int EPI>
__global__ __launch_bounds__(CS_BLOCK) void csr_cusort_kernel(CuSortView A, const real *__restrict__ x, real *y, EpiArgs e, const int *skip) {
  if (skip && *skip) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_smem[];
  __shared__ real red[CS_WAVES];
  real *acc = reinterpret_cast<real *>(cs_smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const unsigned cmask = (1u << A.cbits) - 1;
  real dot = 0;
  for (int u = blockIdx.x; u < A.nunit; u += gridDim.x) {
    const int r0 = A.urow[u], nr = A.urow[u + 1] - r0;
    const int c0 = A.uchunk[u], c1 = A.uchunk[u + 1];
    for (int k = tid; k < nr; k += CS_BLOCK) acc[k] = 0;
    for (int c = c0; c < c1; ++c) {
      const size_t off = (size_t)c * CS_CHUNK + (size_t)tid * 4;
      const uint4 w4 = *reinterpret_cast<const uint4 *>(A.wrd + off);
      const double2 va = *reinterpret_cast<const double2 *>(A.val + off), vb = *reinterpret_cast<const double2 *>(A.val + off + 2);
      const int base = A.cbase[c];
      const unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
      const real v[4] = {(real)va.x, (real)va.y, (real)vb.x, (real)vb.y};
      real xx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __syncthreads(); // the 16 waves issue round i's gathers together (and, for i == 0, the previous chunk's last adds are complete)
        xx[i] = x[base + (int)(w[i] & cmask)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i > 0) __syncthreads(); // round i - 1's adds of every wave are complete before any add of round i
        lds_add(acc + (w[i] >> A.cbits), v[i] * xx[i]);
      }
    }
    __syncthreads();
    for (int k = tid; k < nr; k += CS_BLOCK) {
      const real a = epi_init<EPI>(e, y, r0 + k) + acc[k];
      epi_apply<EPI>(e, y, r0 + k, a, dot);
    }
    __syncthreads();
  }
  if (EPI == EPI_GP && e.partial) {
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    if (tid == 0) {
      real sum = red[0];
      for (int i = 1; i < CS_WAVES; ++i) sum += red[i];
      e.partial[blockIdx.x] = sum;
    }
  }
}
#endif // __HIPCC__

struct CuSortDev {
  bool built = false;
  int rows = 0, cols = 0, nunit = 0, cbits = 0, rows_max = 0;
  long long nchunk = 0, deferred = 0, slots = 0;
  DevBuf<int> urow, uchunk, cbase;
  DevBuf<unsigned> wrd;
  DevBuf<real> val;
  // host copies kept only when the caller asks for an emulation (tests)
  std::vector<int> h_urow, h_uchunk, h_cbase;
  std::vector<unsigned> h_wrd;
  std::vector<real> h_val;
  CuSortView view() const { return CuSortView{rows, nunit, cbits, urow.p, uchunk.p, cbase.p, wrd.p, val.p}; }
  int grid() const { return std::max(1, nunit); }
  size_t lds_bytes() const { return (size_t)rows_max * sizeof(real); }

  // returns false (nothing allocated) when the packed word cannot hold a chunk's column span next to the local row
  bool build_host(int rows_, int cols_, const int *hptr, const int *hidx, const real *hval, int cus) {
    rows = rows_;
    cols = cols_;
    const long long nnz_all = hptr[rows];
    if (rows < 1 || nnz_all < 1) return false;
    // ---- units: one per CU, balanced by nonzeros
    long long budget = std::max<long long>(1, (nnz_all + cus - 1) / cus);
    for (int tries = 0; tries < 64; ++tries) { // greedy packing overshoots by a unit or two: widen until one unit per CU (a tiny tail unit's
      h_urow.assign(1, 0);                     // single chunk would span every column)
      for (int r = 0; r < rows;) {
        const int s0 = r;
        long long acc = 0;
        while (r < rows && r - s0 < CS_ROWS_MAX) {
          const long long rn = hptr[r + 1] - hptr[r];
          if (acc + rn > budget && r > s0) break;
          acc += rn;
          ++r;
        }
        h_urow.push_back(r);
      }
      if ((int)h_urow.size() - 1 <= cus || (long long)rows > (long long)CS_ROWS_MAX * cus) break;
      budget += std::max<long long>(1, budget / 200);
    }
    nunit = (int)h_urow.size() - 1;
    rows_max = 2;
    for (int u = 0; u < nunit; ++u) rows_max = std::max(rows_max, h_urow[u + 1] - h_urow[u]);
    int rbits = 1;
    while ((1 << rbits) < rows_max) ++rbits;
    // ---- per unit: sort by column, deal into rounds, pack
    struct Ent {
      int col, row;
      real v;
    };
    std::vector<Ent> ent, carry, next_carry;
    std::vector<int> stamp((size_t)rows_max, -1), wave_of((size_t)rows_max, 0);
    std::vector<int> cnt;
    h_uchunk.assign(1, 0);
    h_cbase.clear();
    std::vector<Ent> slot; // the unit's slots in (round, q) order, padding = row -1
    std::vector<Ent> all_slots;
    std::vector<long long> unit_slot0;
    int span_max = 1;
    long long round_id = 0;
    deferred = 0;
    std::vector<std::vector<Ent>> unit_slots((size_t)nunit);
    for (int u = 0; u < nunit; ++u) {
      const int r0 = h_urow[u], r1 = h_urow[u + 1];
      const int k0 = hptr[r0], k1 = hptr[r1];
      // counting sort by column bucket, then insertion inside the (short) buckets: row-major input order breaks ties by row
      const int nb = 8192;
      const long long width = ((long long)cols + nb - 1) / nb;
      cnt.assign((size_t)nb + 1, 0);
      for (int k = k0; k < k1; ++k) cnt[(size_t)(hidx[k] / width) + 1]++;
      for (int b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
      ent.resize((size_t)(k1 - k0));
      {
        std::vector<int> fill(cnt.begin(), cnt.end() - 1);
        for (int r = r0; r < r1; ++r)
          for (int k = hptr[r]; k < hptr[r + 1]; ++k) ent[(size_t)fill[hidx[k] / width]++] = Ent{hidx[k], r - r0, hval[k]};
      }
      for (int b = 0; b < nb; ++b)
        std::stable_sort(ent.begin() + cnt[b], ent.begin() + cnt[b + 1], [](const Ent &a, const Ent &c) { return a.col < c.col; });
      // deal into rounds of CS_ROUND slots; an entry whose row already sits in ANOTHER wave of the round waits for the next round
      std::vector<Ent> &out = unit_slots[u];
      out.clear();
      carry.clear();
      size_t pos = 0;
      while (pos < ent.size() || !carry.empty()) {
        const size_t round0 = out.size();
        next_carry.clear();
        ++round_id;
        auto place = [&](const Ent &en) -> bool { // false: the round is full
          const size_t filled = out.size() - round0;
          if (filled >= (size_t)CS_ROUND) return false;
          const int w = (int)(filled / 64);
          if (stamp[en.row] == (int)(round_id & 0x7fffffff) && wave_of[en.row] != w) {
            next_carry.push_back(en);
            ++deferred;
            return true;
          }
          stamp[en.row] = (int)(round_id & 0x7fffffff);
          wave_of[en.row] = w;
          out.push_back(en);
          return true;
        };
        size_t ci = 0;
        for (; ci < carry.size(); ++ci)
          if (!place(carry[ci])) break;
        for (; ci < carry.size(); ++ci) next_carry.push_back(carry[ci]); // (round filled up by carried entries alone: cannot happen in practice)
        while (pos < ent.size() && place(ent[pos])) ++pos;
        while ((out.size() - round0) < (size_t)CS_ROUND) out.push_back(Ent{-1, 0, (real)0}); // padding
        carry.swap(next_carry);
      }
      while (out.size() % CS_CHUNK) out.push_back(Ent{-1, 0, (real)0});
      const int nch = (int)(out.size() / CS_CHUNK);
      for (int c = 0; c < nch; ++c) {
        int lo = cols, hi = -1;
        for (int q = 0; q < CS_CHUNK; ++q) {
          const Ent &en = out[(size_t)c * CS_CHUNK + q];
          if (en.col >= 0) {
            lo = std::min(lo, en.col);
            hi = std::max(hi, en.col);
          }
        }
        if (hi < 0) lo = hi = 0;
        h_cbase.push_back(lo);
        span_max = std::max(span_max, hi - lo + 1);
      }
      h_uchunk.push_back(h_uchunk.back() + nch);
    }
    cbits = 1;
    while ((1ll << cbits) < span_max) ++cbits;
    if (getenv("SCS_AMD_DEBUG"))
      fprintf(stderr, "[scs_amd cusort] rows_max %d (%d bits), largest column span of a chunk %d (%d bits), %lld deferred\n", rows_max, rbits, span_max, cbits, deferred);
    if (cbits + rbits > 32) return false;
    nchunk = h_uchunk.back();
    slots = nchunk * CS_CHUNK;
    if (slots >= (1ll << 31)) return false;
    h_wrd.assign((size_t)slots, 0u);
    h_val.assign((size_t)slots, (real)0);
    for (int u = 0; u < nunit; ++u) {
      const std::vector<Ent> &out = unit_slots[u];
      const int nch = (int)(out.size() / CS_CHUNK);
      for (int c = 0; c < nch; ++c) {
        const int base = h_cbase[(size_t)h_uchunk[u] + c];
        for (int i = 0; i < 4; ++i)
          for (int q = 0; q < CS_ROUND; ++q) {
            const Ent &en = out[(size_t)c * CS_CHUNK + (size_t)i * CS_ROUND + q];
            const size_t p = ((size_t)h_uchunk[u] + c) * CS_CHUNK + (size_t)q * 4 + i;
            if (en.col >= 0) {
              h_wrd[p] = (unsigned)(en.col - base) | ((unsigned)en.row << cbits);
              h_val[p] = en.v;
            }
          }
      }
      std::vector<Ent>().swap(unit_slots[u]);
    }
    return true;
  }

  void upload(hipStream_t st, bool keep_host) {
    urow.alloc(h_urow.size());
    uchunk.alloc(h_uchunk.size());
    cbase.alloc(std::max<size_t>(1, h_cbase.size()));
    wrd.alloc((size_t)slots + 8);
    val.alloc((size_t)slots + 8);
    urow.upload(h_urow.data(), h_urow.size(), st);
    uchunk.upload(h_uchunk.data(), h_uchunk.size(), st);
    if (!h_cbase.empty()) cbase.upload(h_cbase.data(), h_cbase.size(), st);
    wrd.upload(h_wrd.data(), (size_t)slots, st);
    val.upload(h_val.data(), (size_t)slots, st);
    HIP_CHECK(hipStreamSynchronize(st));
    if (!keep_host) {
      std::vector<unsigned>().swap(h_wrd);
      std::vector<real>().swap(h_val);
      std::vector<int>().swap(h_cbase);
    }
    built = true;
  }

  // the kernel's arithmetic on the host, in the kernel's order (tests: the layout and the dealing are right without a GPU)
  void emulate(const real *x, real *y) const {
    std::vector<real> acc((size_t)rows_max);
    const unsigned cmask = (1u << cbits) - 1;
    for (int u = 0; u < nunit; ++u) {
      const int r0 = h_urow[u], nr = h_urow[u + 1] - r0;
      std::fill(acc.begin(), acc.begin() + nr, (real)0);
      for (int c = h_uchunk[u]; c < h_uchunk[u + 1]; ++c)
        for (int i = 0; i < 4; ++i)
          for (int q = 0; q < CS_ROUND; ++q) {
            const size_t p = (size_t)c * CS_CHUNK + (size_t)q * 4 + i;
            acc[h_wrd[p] >> cbits] += h_val[p] * x[h_cbase[c] + (int)(h_wrd[p] & cmask)];
          }
      for (int k = 0; k < nr; ++k) y[r0 + k] = acc[k];
    }
  }
};

} // namespace scsamd
