// spmv_wave.h -- "wave-owned rows" CSR product for matrices from a million nonzeros on
// (same results and epilogues as csr_stream_kernel in spmv.h; replaces round 1's column-sliced
// kernel).  Replaces SCS(accum_by_atrans), reference linsys/scs_matrix.c:161-186.
//
// Why: on the headline config the product is bound by the scattered 8-byte reads of x, each of
// which moves a whole 128-byte line from L2 to the CU (lab/spmv_lab.hip: 1e7 such reads take
// 37 us L2-resident, 80 us from an 8 MB table, 111 us from 16 MB; profiles/r2_spmv_lab.md), and
// round 1's kernel added a load -> gather -> LDS -> barrier -> segmented sum -> barrier chain per
// 512 products on top of that.
//
// How: a *wave* owns a unit of <= 1024 consecutive rows and keeps one accumulator per row in a
// private slice of LDS.  It streams the unit's entries -- val (8 B) + one packed word
// (column | local row << cbits) = the same 12 B/nnz as CSR and no row pointers -- with unit
// stride, gathers x[column] and adds val * x to its accumulator with an LDS atomic
// (ds_add_f64 / ds_add_f32).  A wave's LDS operations execute in program order and nothing else
// touches its accumulators, so there is no barrier, no product staging and no sort in the
// kernel, and the sum is reproducible bit for bit.  Inside a unit the entries are stored by
// ascending column bucket; all waves are resident at once and start together, so at any moment
// the whole chip gathers from one window of x that moves upwards and stays in every XCD's L2
// (measured 91 % L2 hits on the gathers, profiles/r2_g2_lab.md section 7).  Units are balanced by
// nonzeros (about 8 waves per CU: fewer, fatter waves measured faster than many thin ones).
// Summation order inside a row is column-bucket order with row-major ties (rounding-level
// difference from the reference's index order).
// Measured (lab/g2_lab.hip, headline shapes): 54-56 us (A) / 59 us (A') vs 62-65 us for the
// sliced kernel in the same harness.
#pragma once
#include "spmv.h"
#include <algorithm>

namespace scsamd {

constexpr int WR_WPB = 4;           // waves per workgroup (one unit each)
constexpr int WR_BLOCK = WR_WPB * 64;
constexpr int WR_ROWS_MAX = 1024;   // rows per unit (8 KB of fp64 accumulators per wave)
constexpr int WR_MAX_GRID = 4096;   // partial-array bound for the fused dot product
constexpr int WR_BUCKETS = 1024;    // column buckets per unit (ordering heuristic only)

struct WaveView {
  int rows, nunit, cbits, cols;
  const int *urow;     // nunit + 1 : first row of each unit
  const eoff *useg;    // 2 * nunit : [first entry (4-aligned), one past the last entry] of each unit (entry positions: eoff)
  const unsigned *wrd; // nnz : column | local row << cbits  (wide layout: the column alone)
  const real *val;     // nnz
  const unsigned short *rowl; // wide layout only (gathered vectors beyond 2^26 entries, round 6): the local row of every entry; else null
};

#ifdef __HIPCC__
__device__ __forceinline__ void lds_add(real *p, real v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one lane's 4 consecutive entries of a 256-entry chunk: one 16-byte load of packed words, 16-byte loads of values
struct WrChunk {
  uint4 w;
  real v[4];
};
__device__ __forceinline__ WrChunk wr_load(const WaveView &A, eoff eb) {
  WrChunk c;
  c.w = *reinterpret_cast<const uint4 *>(A.wrd + eb);
  if (sizeof(real) == 8) {
    const double2 va = *reinterpret_cast<const double2 *>(A.val + eb), vb = *reinterpret_cast<const double2 *>(A.val + eb + 2);
    c.v[0] = (real)va.x; c.v[1] = (real)va.y; c.v[2] = (real)vb.x; c.v[3] = (real)vb.y;
  } else {
    const float4 va = *reinterpret_cast<const float4 *>(A.val + eb);
    c.v[0] = (real)va.x; c.v[1] = (real)va.y; c.v[2] = (real)va.z; c.v[3] = (real)va.w;
  }
  return c;
}
__device__ __forceinline__ void wr_consume(const WaveView &A, const WrChunk &c, const real *__restrict__ x, real *acc, eoff eb, eoff t,
                                           unsigned cmask) {
  const unsigned w[4] = {c.w.x, c.w.y, c.w.z, c.w.w};
  real xx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xx[i] = eb + i < t ? x[w[i] & cmask] : (real)0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (eb + i < t) lds_add(acc + (w[i] >> A.cbits), c.v[i] * xx[i]);
}

// one partial per workgroup and dot product: waves in a fixed order (bit-reproducible)
template <int EPI, int WPB>
__device__ __forceinline__ void wr_block_partials(const EpiArgs &e, real (*red)[WPB], real dot, real d1, real d2, int wave, int lane) {
  dot = wave_sum(dot);
  if (EPI == EPI_GP3) {
    d1 = wave_sum(d1);
    d2 = wave_sum(d2);
  }
  if (lane == 0) {
    red[0][wave] = dot;
    if (EPI == EPI_GP3) {
      red[1][wave] = d1;
      red[2][wave] = d2;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    real sum = red[0][0];
    for (int i = 1; i < WPB; ++i) sum += red[0][i];
    e.partial[blockIdx.x] = sum;
    if (EPI == EPI_GP3) {
      real s1 = red[1][0], s2 = red[2][0];
      for (int i = 1; i < WPB; ++i) {
        s1 += red[1][i];
        s2 += red[2][i];
      }
      e.partial2[blockIdx.x] = s1;
      e.partial3[blockIdx.x] = s2;
    }
  }
}

// (Round 4, measured and not kept: an "x-window prefetch" instantiation -- every wave loads its share of the window of x the chip will
// gather from 1..6 chunks later, next to its stream loads, so that the gathers hit L2 instead of pulling each line of x out of the
// Infinity Cache once per XCD: 70.9 us per product at a look-ahead of 1 chunk, 77.5 - 79.8 us at 2 - 6, against 68.2 - 68.8 us
// without; profiles/r4_xwindow_prefetch.md.  One more far-latency load per chunk in the CU's in-order queue costs more than the
// L2 misses of the gathers it removes.)
// PIPE = 1: the stream loads of the next chunk are in flight while the current chunk gathers and accumulates.  Chosen
// per matrix (WaveRowsDev::pipelined) from the measured line sharing of its gathers: with column locality the gathers are
// L1 / L2 hits, the product is the 12 B/nnz stream and wants more bytes in flight (headline sizes, columns confined to a
// band of 1024 / 4096 / 65536 rows: 45.8 -> 39.8, 51.3 -> 46.3, 68.8 -> 65.3 us per product; profiles/r3_locality_v2.jsonl);
// when every gather is its own line fill (the uniformly random matrices of the headline benchmark) there is nothing to
// win (69.1 vs 69.6 us), and two chunks ahead lose everywhere (42.4 / 51.7 / 70.0 us; profiles/r2_g4_lab.md for the
// random case: more HBM loads queue ahead of the gathers in the CU's in-order memory path).
template <int EPI, int PIPE>
__global__ __launch_bounds__(WR_BLOCK) void csr_wave_kernel(WaveView A, const real *__restrict__ x, real *y, EpiArgs e,
                                                            const int *skip, int accrows) {
  if (skip && *skip) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char wr_smem[];
  __shared__ real red[3][WR_WPB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  real *acc = reinterpret_cast<real *>(wr_smem) + (size_t)wave * accrows;
  const unsigned cmask = (1u << A.cbits) - 1;
  real dot = 0, d1 = 0, d2 = 0;
  for (int u = blockIdx.x * WR_WPB + wave; u < A.nunit; u += gridDim.x * WR_WPB) {
    const int r0 = A.urow[u], nr = A.urow[u + 1] - r0;
    const eoff s = A.useg[2 * u], t = A.useg[2 * u + 1];
    for (int k = lane; k < nr; k += 64) acc[k] = 0;
    // a lane owns 4 consecutive entries of every 256-entry chunk (3 stream instructions per chunk instead of 8; unit
    // starts are 4-aligned)
    if (PIPE == 1) {
      WrChunk cur = wr_load(A, s + lane * 4); // (an empty unit reads the padding behind its start: harmless)
      for (eoff e0 = s; e0 < t; e0 += 256) {
        const bool more = e0 + 256 < t; // uniform
        WrChunk nxt = cur;
        if (more) nxt = wr_load(A, e0 + 256 + lane * 4);
        wr_consume(A, cur, x, acc, e0 + lane * 4, t, cmask);
        cur = nxt;
      }
    } else if (PIPE == 2) {
      // round 6, mid-size systems (the batch config's n = 2e5: units of ~4 chunks, the gathered vector fits an XCD's L2, the matrix the
      // Infinity Cache): the product is a chain of dependent round trips per wave -- stream, gathers, adds, once per chunk -- not a
      // bandwidth problem.  TWO chunks per trip: both chunks' stream loads are issued together, then all eight gather instructions, then
      // the adds in the plain kernel's order (chunk c before chunk c + 1, entry i before i + 1: the same bits as PIPE = 0).
      for (eoff e0 = s; e0 < t; e0 += 512) {
        const eoff ea = e0 + lane * 4, eb = ea + 256;
        const bool two = e0 + 256 < t; // uniform (the padding behind a unit covers ONE chunk of over-read, not two)
        const WrChunk c0 = wr_load(A, ea);
        WrChunk c1 = c0;
        if (two) c1 = wr_load(A, eb);
        const unsigned w0[4] = {c0.w.x, c0.w.y, c0.w.z, c0.w.w}, w1[4] = {c1.w.x, c1.w.y, c1.w.z, c1.w.w};
        real x0[4], x1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x0[i] = ea + i < t ? x[w0[i] & cmask] : (real)0;
#pragma unroll
        for (int i = 0; i < 4; ++i) x1[i] = (two && eb + i < t) ? x[w1[i] & cmask] : (real)0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (ea + i < t) lds_add(acc + (w0[i] >> A.cbits), c0.v[i] * x0[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (two && eb + i < t) lds_add(acc + (w1[i] >> A.cbits), c1.v[i] * x1[i]);
      }
    } else {
      for (eoff e0 = s; e0 < t; e0 += 256) {
        const eoff eb = e0 + lane * 4;
        const WrChunk c = wr_load(A, eb);
        wr_consume(A, c, x, acc, eb, t, cmask);
      }
    }
    for (int k = lane; k < nr; k += 64) {
      const real a = epi_init<EPI>(e, y, r0 + k) + acc[k];
      if (EPI == EPI_GP3) epi_apply3(e, y, r0 + k, a, dot, d1, d2);
      else epi_apply<EPI>(e, y, r0 + k, a, dot);
    }
  }
  if ((EPI == EPI_GP || EPI == EPI_GP3) && e.partial) wr_block_partials<EPI, WR_WPB>(e, red, dot, d1, d2, wave, lane);
}
// WIDE layout (round 6): a gathered vector of more than 2^26 entries leaves no room for the local row in the 32-bit packed word (the run with
// nnz = 2.2e9, n = 1e8, m = 2e8 fell back to csr_stream_kernel at 0.085 of the byte roofline, profiles/r6_dlong_real.json -- and so did
// this layout when it was built: 0.086; see WaveRowsDev::wanted).
// The word then holds the column alone and the local row travels in a 16-bit array of its own: 14 B per entry instead of 12, the same
// wave-owned rows, LDS accumulators, bucket order and epilogues.  The plain schedule only (PIPE = 0; no lockstep): the same entry order and
// the same sums, bit for bit, as csr_wave_kernel<EPI, 0> on the narrow layout of the same matrix (tests force it at small sizes, option wr_wide).
template <int EPI>
__global__ __launch_bounds__(WR_BLOCK) void csr_wave_wide_kernel(WaveView A, const real *__restrict__ x, real *y, EpiArgs e, const int *skip,
                                                                 int accrows) {
  if (skip && *skip) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char wr_smem[];
  __shared__ real red[3][WR_WPB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  real *acc = reinterpret_cast<real *>(wr_smem) + (size_t)wave * accrows;
  real dot = 0, d1 = 0, d2 = 0;
  for (int u = blockIdx.x * WR_WPB + wave; u < A.nunit; u += gridDim.x * WR_WPB) {
    const int r0 = A.urow[u], nr = A.urow[u + 1] - r0;
    const eoff s = A.useg[2 * u], t = A.useg[2 * u + 1];
    for (int k = lane; k < nr; k += 64) acc[k] = 0;
    for (eoff e0 = s; e0 < t; e0 += 256) {
      const eoff eb = e0 + lane * 4;
      const WrChunk c = wr_load(A, eb);
      const uint2 rr = *reinterpret_cast<const uint2 *>(A.rowl + eb); // four 16-bit local rows (unit starts are 4-aligned: 8-byte aligned)
      const unsigned w[4] = {c.w.x, c.w.y, c.w.z, c.w.w};
      const unsigned lr[4] = {rr.x & 0xffffu, rr.x >> 16, rr.y & 0xffffu, rr.y >> 16};
      real xx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xx[i] = eb + i < t ? x[w[i]] : (real)0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (eb + i < t) lds_add(acc + lr[i], c.v[i] * xx[i]);
    }
    for (int k = lane; k < nr; k += 64) {
      const real a = epi_init<EPI>(e, y, r0 + k) + acc[k];
      if (EPI == EPI_GP3) epi_apply3(e, y, r0 + k, a, dot, d1, d2);
      else epi_apply<EPI>(e, y, r0 + k, a, dot);
    }
  }
  if ((EPI == EPI_GP || EPI == EPI_GP3) && e.partial) wr_block_partials<EPI, WR_WPB>(e, red, dot, d1, d2, wave, lane);
}
// Lockstep instantiation (round 4; the library's choice for fp64 systems from 5e6 nonzeros on, SCS_AMD_WR_LOCKSTEP = 0 | 1 forces either): ONE
// workgroup of 16 waves per CU, one unit per wave (units half as long as the plain kernel's).  The host stores every 256-entry chunk so that
// gather instruction i of a wave covers the i-th QUARTER of the chunk's column window (rank r of the chunk's entries in column order sits at
// position 4 (r % 64) + r / 64), and a workgroup barrier in front of every gather instruction makes the 16 waves of the CU issue gather
// instruction i of their chunk c together.  What that buys, measured (profiles/r4_spmv_lockstep.md, r4_pmc_l1*.md): 69.1 -> 62.2 us per product
// on the headline (0.27 -> 0.30 of 8 TB/s), 39.0 -> 34.0 us on a band of 1024 rows (0.48 -> 0.55), 64.8 -> 51.3 us on a band of 65536.  NOT
// mainly through L1 reuse -- the L1 -> L2 read requests per product fall by 3 % only (1.052e7 -> 1.016e7 / 1.085e7 -> 1.067e7) -- but through
// the order of the CU's in-order vector-memory queue: with every wave of the CU in the same phase, a batch of L2-hit gathers is no longer
// interleaved with other waves' HBM stream loads (TCP_PENDING_STALL_CYCLES -8 %), and sixteen waves' worth of gathers are in flight without
// the drift that made sixteen UN-synchronised waves per CU slower than eight (91 vs 69 us, round 3).  Same per-wave LDS accumulators and
// program-order adds as the plain kernel: bit-reproducible.  Variants measured and dropped: 8 waves per workgroup (66.4 us), one barrier per
// chunk (65.1; kept for matrices whose gathers share lines: 34.0 vs 35.3 us on the band), the next chunk's stream loads issued behind the
// gathers (67.3) or ahead of them (62.6), an x-window prefetch (73.7), 16 waves as two workgroups (69.7).
template <int EPI, int WL_WPB, int MODE> // MODE = barriers per chunk (4: in front of every gather instruction | 1)
__global__ __launch_bounds__(WL_WPB * 64) void csr_wave_lockstep_kernel(WaveView A, const real *__restrict__ x, real *y, EpiArgs e,
                                                                        const int *skip, int accrows) {
  if (skip && *skip) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char wr_smem[];
  __shared__ real red[3][WL_WPB];
  __shared__ int s_nch[WL_WPB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  real *acc = reinterpret_cast<real *>(wr_smem) + (size_t)wave * accrows;
  const unsigned cmask = (1u << A.cbits) - 1;
  real dot = 0, d1 = 0, d2 = 0;
  const int per_round = gridDim.x * WL_WPB;
  const int nround = (A.nunit + per_round - 1) / per_round;
  for (int rd = 0; rd < nround; ++rd) {
    const int u = rd * per_round + blockIdx.x * WL_WPB + wave;
    const bool live = u < A.nunit; // wave-uniform
    int r0 = 0, nr = 0;
    eoff s = 0, t = 0;
    if (live) {
      r0 = A.urow[u];
      nr = A.urow[u + 1] - r0;
      s = A.useg[2 * u];
      t = A.useg[2 * u + 1];
    }
    for (int k = lane; k < nr; k += 64) acc[k] = 0;
    const int nch = (int)((t - s + 255) >> 8);
    if (lane == 0) s_nch[wave] = nch;
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int w = 0; w < WL_WPB; ++w) nmax = s_nch[w] > nmax ? s_nch[w] : nmax;
    constexpr int bars = MODE; // barriers per chunk: 4 | 1
    auto consume = [&](const WrChunk &ch, int c) {
      const eoff eb = s + c * 256 + lane * 4;
      const bool has = c < nch; // wave-uniform: a wave whose unit is shorter keeps the others company at the barriers
      const unsigned w[4] = {ch.w.x, ch.w.y, ch.w.z, ch.w.w};
      real xx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (bars == 4 || (bars == 2 && (i & 1) == 0) || (bars == 1 && i == 0))
          __syncthreads(); // all waves of the CU issue gather instruction i of their chunk c together (bars: barriers per chunk)
        xx[i] = (has && eb + i < t) ? x[w[i] & cmask] : (real)0;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (has && eb + i < t) lds_add(acc + (w[i] >> A.cbits), ch.v[i] * xx[i]);
    };
    WrChunk zero;
    zero.w = make_uint4(0, 0, 0, 0);
    zero.v[0] = zero.v[1] = zero.v[2] = zero.v[3] = 0;
    for (int c = 0; c < nmax; ++c) {
      const WrChunk ch = c < nch ? wr_load(A, s + c * 256 + lane * 4) : zero;
      consume(ch, c);
    }
    for (int k = lane; k < nr; k += 64) {
      const real a = epi_init<EPI>(e, y, r0 + k) + acc[k];
      if (EPI == EPI_GP3) epi_apply3(e, y, r0 + k, a, dot, d1, d2);
      else epi_apply<EPI>(e, y, r0 + k, a, dot);
    }
    __syncthreads(); // s_nch is rewritten by the next round
  }
  if ((EPI == EPI_GP || EPI == EPI_GP3) && e.partial) wr_block_partials<EPI, WL_WPB>(e, red, dot, d1, d2, wave, lane);
}
#endif // __HIPCC__

struct WaveRowsDev {
  bool built = false;
  bool built_on_device = false; // spmv_wave_build.h filled wrd / val (else fill_host did)
  int pipelined = 0;           // 1: one chunk of stream in flight ahead of the gathers (csr_wave_kernel<.., 1>): matrices whose gathers share lines
  double lines_per_entry = 1;  // distinct 128-byte lines of x a unit touches / its entries, averaged (1 = every gather its own line)
  int rows = 0, cols = 0, nunit = 0, cbits = 0, accrows = 0, cus = 256;
  int lockstep = 0;            // 1: csr_wave_lockstep_kernel (8 or 16 waves per workgroup, sub-window chunk order; SCS_AMD_WR_LOCKSTEP)
  int sub_window_order = 0;    // every 256-entry chunk stored so that gather instruction i covers the i-th quarter of its column window
  int ls_wpb = 16, ls_bmode = 4; // its waves per workgroup (SCS_AMD_WR_LS_WPB = 8 | 16) and barriers per chunk (SCS_AMD_WR_LS_BARRIERS = 4 | 1)
  int wpc = 8;                 // waves per CU the layout is cut for (one unit per resident wave); SCS_AMD_WR_WPC overrides (measurements)
  DevBuf<int> urow;
  DevBuf<eoff> useg;
  long long bias = 0; // test hook of the DLONG build (CsrDev::bias): entry positions stored +bias, arrays handed over shifted by -bias
  DevBuf<unsigned> wrd;
  DevBuf<real> val;
  bool wide = false;            // gathered vector beyond 2^26 entries (or option wr_wide): column word + 16-bit local rows (csr_wave_wide_kernel)
  DevBuf<unsigned short> rowl;  // wide layout only
  WaveView view() const { return WaveView{rows, nunit, cbits, cols, urow.p, useg.p, wrd.p - bias, val.p - bias, wide ? rowl.p - bias : nullptr}; }
  // distinct 128-byte lines per unit are COUNTED (a bitmap over the gathered vector's lines, in LDS on the device) only while that bitmap is
  // small; beyond, both builders report one line per entry (the count only picks the stream flavour)
  static bool lines_counted(int cols) { return ((long long)cols >> (sizeof(real) == 8 ? 4 : 5)) / 8 <= 64 * 1024; }
  // every workgroup must be resident at once (8 waves per CU): a wave then walks its units one after the
  // other and all waves restart at column 0 together, which keeps the gather window of x aligned; a second
  // generation of workgroups starting at column 0 while the first is half way through thrashes L2 instead
  // (measured at n = 4e6: 415 us per product with 1954 workgroups vs the resident grid)
  // equal rounds: with R = ceil(nunit / (8 cus)) rounds, ceil(nunit / R) units run at a time
  int grid() const {
    const int resident = wpc * cus;
    const int rounds = std::max(1, (nunit + resident - 1) / resident);
    const int per_round = (nunit + rounds - 1) / rounds;
    const int wpb = lockstep ? ls_wpb : WR_WPB;
    return std::max(1, std::min((per_round + wpb - 1) / wpb, WR_MAX_GRID));
  }
  size_t lds_bytes() const { return (size_t)(lockstep ? ls_wpb : WR_WPB) * accrows * sizeof(real); }
  static int col_bits(int cols) {
    int b = 1;
    while ((1ll << b) < cols) ++b;
    return b;
  }
  // From a million nonzeros on: measured us per CG iteration, this kernel vs csr_stream (fp64, 10 nonzeros per
  // column): nnz 5e5 23.5 / 23.5, 1e6 30.0 / 32.7, 1.5e6 34.8 / 38.5, 2e6 39.3 / 43.3, 4e6 61.3 / 76.5 -- no
  // barriers and no product staging pay even while the gathered vector still fits an XCD's L2
  static bool wanted(int cols, const eoff *hptr, int rows) {
    // Beyond 26 column bits the packed word has no room for the local row.  The WIDE layout (round 6) handles that, but is NOT chosen by
    // default: measured on the nnz = 2.2e9 problem (n = 1e8, m = 2e8) it times exactly like the CSR-stream kernel -- 42.9 vs 43.0 ms per
    // product, both at one HBM line per gather (2.2e9 x 128 B = 282 GB per product at 6.7 TB/s: at that size no window of the gathered
    // vector stays in any cache) -- for 61 GB more HBM (profiles/r6_dlong_real.json).  Option wr_wide = 1 selects it.
    if (col_bits(cols) > 26) {
      const char *w = opt_get("wr_wide");
      if (!w || !atoi(w)) return false;
    }
    if (const char *e = opt_get("waverows")) return atoi(e) != 0; // tests force either path
    return (long long)hptr[rows] >= 1000000LL;
  }
  // ---- layout construction, in three parts (round 5):
  //   plan       (host, O(rows)):  which kernel flavour, the unit partition by nonzeros, entry offsets of the units
  //   fill_host  (host, O(nnz)):   the entries of every unit ordered by column bucket (+ the quarter-window chunk order of the lockstep
  //                                kernel), distinct lines per entry -- rounds 2-4's builder, kept as the general path (units longer
  //                                than WR_DEV_UNIT_MAX entries) and as the oracle of the device builder (SCS_AMD_WR_BUILD=verify)
  //   fill_dev   (device):         the same arrays, bit for bit, from the CSR copy already in HBM (spmv_wave_build.h): one workgroup
  //                                per unit, two in-LDS bitonic sorts on composite keys that reproduce the host's stable orders
  std::vector<int> ur;   // host copy of urow (plan)
  std::vector<eoff> us;  // host copy of useg
  size_t cap = 0;
  int bshift = 0;
  long long nnz_all = 0;
  void plan(int rows_, int cols_, const eoff *hptr) {
    rows = rows_;
    cols = cols_;
    cbits = col_bits(cols);
    wide = cbits > 26;
    if (const char *e = opt_get("wr_wide")) wide = wide || atoi(e) != 0; // tests force the wide layout at small sizes
    const int rows_cap = wide ? WR_ROWS_MAX : (int)std::min<long long>(WR_ROWS_MAX, 1ll << (32 - cbits));
    nnz_all = hptr[rows];
    // nonzero budget per unit: ~8 waves per CU on the whole chip (SCS_AMD_WR_NNZ overrides)
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
      cus = 256;
    // Lockstep instantiation (csr_wave_lockstep_kernel: one workgroup of 16 waves per CU, gathers issued together): measured faster for
    // fp64 from ~5e6 nonzeros on, slower below and in fp32 (profiles/r4_spmv_lockstep.md); SCS_AMD_WR_LOCKSTEP = 0 | 1 forces either,
    // 2 = its chunk order with the plain kernel
    lockstep = (sizeof(real) == 8 && nnz_all >= 5000000LL) ? 1 : 0;
    sub_window_order = lockstep;
    ls_wpb = 16;
    ls_bmode = -1; // chosen below from the measured line sharing unless SCS_AMD_WR_LS_BARRIERS says otherwise
    if (const char *e = opt_get("wr_lockstep")) {
      lockstep = atoi(e) == 1 ? 1 : 0;
      sub_window_order = atoi(e) != 0;
    }
    if (const char *w = opt_get("wr_ls_wpb")) ls_wpb = atoi(w) == 8 ? 8 : 16;
    if (const char *b = opt_get("wr_ls_barriers")) ls_bmode = atoi(b) == 1 ? 1 : 4;
    if (const char *o = opt_get("wr_ls_order")) sub_window_order = atoi(o) != 0; // measurements: lockstep without the quarter-window chunk order
    if (wide) lockstep = sub_window_order = 0; // the wide layout runs the plain schedule only
    if (lockstep) wpc = ls_wpb; // one workgroup per CU, one unit per wave
    if (const char *e = opt_get("wr_wpc")) wpc = std::max(1, std::min(32, atoi(e)));
    long long budget = std::max<long long>(1024, (nnz_all + (long long)wpc * cus - 1) / ((long long)wpc * cus));
    // wide layout = huge matrices: units short enough for the device builder's in-LDS sort (8192 entries), whatever the resident wave count
    // (at nnz = 2e9 the host builder would walk every entry on one thread, and the walk of a wave over many units costs nothing there)
    if (wide) budget = std::min<long long>(budget, 8192 - 64);
    if (const char *e = opt_get("wr_nnz")) budget = std::max(64, atoi(e));
    auto partition = [&](long long bud) {
      ur.clear();
      ur.push_back(0);
      int r = 0;
      while (r < rows) {
        const int s0 = r;
        long long acc = 0;
        while (r < rows && r - s0 < rows_cap) {
          const long long rn = hptr[r + 1] - hptr[r];
          if (acc + rn > bud && r > s0) break;
          acc += rn;
          ++r;
        }
        ur.push_back(r);
      }
    };
    partition(budget);
    // greedy packing overshoots the resident wave count by a few units, which would cost a whole extra round
    // (measured: 513 workgroups on 512 slots 86 us vs 71 us): widen the budget until the units fit, unless
    // the row cap (packed word) is what limits them
    if (!opt_get("wr_nnz"))
      for (int tries = 0; (long long)ur.size() - 1 > (long long)wpc * cus && (long long)rows <= (long long)rows_cap * wpc * cus && tries < 60; ++tries) {
        budget += std::max<long long>(1, budget / 100);
        partition(budget);
      }
    nunit = (int)ur.size() - 1;
    us.resize((size_t)2 * nunit);
    accrows = 2;
    size_t q = 0; // unit starts are rounded up to 4 entries (16-byte vector loads); the gaps hold zeros
    for (int u = 0; u < nunit; ++u) {
      q = (q + 3) & ~(size_t)3;
      us[2 * u] = (eoff)q;
      q += (size_t)(hptr[ur[u + 1]] - hptr[ur[u]]);
      us[2 * u + 1] = (eoff)q;
      accrows = std::max(accrows, ur[u + 1] - ur[u]);
    }
    accrows = (accrows + 1) & ~1;
    cap = q + 256 + 8; // the last chunk of a unit may read up to 255 entries past its end
    if (sizeof(eoff) == 4 && cap >= ((size_t)1 << 31)) throw HipError("scs_amd: matrix too large for 32-bit entry offsets (the DLONG build carries 64-bit ones)");
    bshift = std::max(0, cbits - 10);
  }
  long long max_unit_entries() const {
    long long mx = 0;
    for (int u = 0; u < nunit; ++u) mx = std::max<long long>(mx, us[2 * u + 1] - us[2 * u]);
    return mx;
  }
  // the kernel flavour that follows from the measured line sharing (both builders end here)
  void finish(long long distinct) {
    lines_per_entry = nnz_all > 0 ? (double)distinct / (double)nnz_all : 1.0;
    pipelined = lines_per_entry < 0.8 ? 1 : 0;
    if (const char *e = opt_get("wr_pipe")) pipelined = std::max(0, std::min(2, atoi(e))); // tests / measurements force 0 | 1 | 2 (two chunks per trip)
    if (wide) pipelined = 0;
    // lockstep: a barrier in front of every gather instruction when every gather is its own line, one per chunk when a unit's
    // gathers share lines anyway (band of 1024 rows at the headline sizes: 34.0 vs 35.3 us per product; uniformly random: 65.1 vs 62.2)
    if (ls_bmode < 0) ls_bmode = lines_per_entry < 0.3 ? 1 : 4;
    built = true;
  }
  void fill_host(const eoff *hptr, const int *hidx, const real *hval, std::vector<unsigned> &hw, std::vector<real> &hv, long long &distinct_total,
                 std::vector<unsigned short> *hr = nullptr) {
    hw.assign(cap, 0u);
    hv.assign(cap, (real)0);
    if (wide) {
      if (!hr) throw HipError("scs_amd: the wide wave layout needs the local-row array");
      hr->assign(cap, (unsigned short)0);
    }
    // stable counting sort of every unit by column bucket (ordering is a locality heuristic: any
    // order gives the same sums up to rounding)
    std::vector<int> cnt(WR_BUCKETS + 1);
    for (int u = 0; u < nunit; ++u) {
      const eoff k0 = hptr[ur[u]], k1 = hptr[ur[u + 1]];
      std::fill(cnt.begin(), cnt.end(), 0);
      for (eoff k = k0; k < k1; ++k) cnt[(hidx[k] >> bshift) + 1]++;
      for (int b = 0; b < WR_BUCKETS; ++b) cnt[b + 1] += cnt[b];
      const size_t base = (size_t)us[2 * u];
      for (int rr = ur[u]; rr < ur[u + 1]; ++rr)
        for (eoff k = hptr[rr]; k < hptr[rr + 1]; ++k) {
          const size_t qq = base + cnt[hidx[k] >> bshift]++;
          if (wide) {
            hw[qq] = (unsigned)hidx[k];
            (*hr)[qq] = (unsigned short)(rr - ur[u]);
          } else {
            hw[qq] = (unsigned)hidx[k] | ((unsigned)(rr - ur[u]) << cbits);
          }
          hv[qq] = hval[k];
        }
    }
    if (sub_window_order) { // every 256-entry chunk: rank r in column order -> position 4 (r % 64) + r / 64 (see csr_wave_lockstep_kernel)
      std::vector<std::pair<unsigned, int>> key(256);
      std::vector<unsigned> tw(256);
      std::vector<real> tv(256);
      for (int u = 0; u < nunit; ++u) {
        const size_t s0 = (size_t)us[2 * u], t0 = (size_t)us[2 * u + 1];
        for (size_t c0 = s0; c0 < t0; c0 += 256) {
          const int len = (int)std::min<size_t>(256, t0 - c0);
          const unsigned cm = (1u << cbits) - 1;
          for (int q = 0; q < len; ++q) key[q] = std::make_pair(hw[c0 + q] & cm, q);
          std::stable_sort(key.begin(), key.begin() + len);
          // a short last chunk keeps the rule "position p is valid iff p < len": ranks are dealt to the valid positions in the order
          // (i = 0: p = 0, 4, 8, ...), (i = 1: p = 1, 5, ...), ... so that instruction i still sees one contiguous window
          int r = 0;
          for (int i = 0; i < 4; ++i)
            for (int l = 0; l < 64; ++l) {
              const int pos = 4 * l + i;
              if (pos >= len) continue;
              tw[pos] = hw[c0 + key[r].second];
              tv[pos] = hv[c0 + key[r].second];
              ++r;
            }
          for (int q = 0; q < len; ++q) {
            hw[c0 + q] = tw[q];
            hv[c0 + q] = tv[q];
          }
        }
      }
    }
    if (!lines_counted(cols)) {
      distinct_total = nnz_all;
    } else { // column locality: how many distinct lines of the gathered vector does a unit touch per entry?
      const int lshift = sizeof(real) == 8 ? 4 : 5; // 128-byte line = 16 fp64 / 32 fp32 entries
      std::vector<int> stamp(((size_t)cols >> lshift) + 2, -1);
      long long distinct = 0;
      for (int u = 0; u < nunit; ++u)
        for (eoff k = hptr[ur[u]]; k < hptr[ur[u + 1]]; ++k) {
          int &st = stamp[(size_t)hidx[k] >> lshift];
          if (st != u) {
            st = u;
            ++distinct;
          }
        }
      distinct_total = distinct;
    }
  }
  void alloc_and_upload_plan(hipStream_t st) {
    urow.alloc(ur.size());
    useg.alloc(us.size());
    wrd.alloc(cap); // zero-filled: the gaps between units and the tail hold zeros
    val.alloc(cap);
    if (wide) rowl.alloc(cap);
    urow.upload(ur.data(), ur.size(), st);
    useg.upload(us.data(), us.size(), st);
  }
  // rounds 2-4's entry point: everything on the host, then uploaded
  void build(int rows_, int cols_, const eoff *hptr, const int *hidx, const real *hval, hipStream_t st) {
    plan(rows_, cols_, hptr);
    std::vector<unsigned> hw;
    std::vector<real> hv;
    long long distinct = 0;
    std::vector<unsigned short> hr;
    fill_host(hptr, hidx, hval, hw, hv, distinct, &hr);
    alloc_and_upload_plan(st);
    wrd.upload(hw.data(), cap, st);
    val.upload(hv.data(), cap, st);
    if (wide) rowl.upload(hr.data(), cap, st);
    HIP_CHECK(hipStreamSynchronize(st));
    finish(distinct);
  }
};

} // namespace scsamd
