// cones.hip -- cone projections as HIP kernels + the B1' host-pointer C ABI.
//
// Reference behaviour restated (src/cones.c):
//   proj_dual_cone        :1552-1596  Moreau wrapper in the R_y metric
//   proj_cone             :1340-1394  dispatcher, order zero,pos,box,SOC,PSD
//   proj_box_cone         :1182-1245  Newton on t (<= 25 its), then clip
//   normalize_box_cone    :1161-1177  bounds scaled by D[j+1]/D[0], +-1e15 -> inf
//   proj_soc              :1250-1279  s = |x_1:|, cases s<=v, s<=-v, else scale
//   proj_semi_definite_cone:999-1067  unpack, sqrt2 on the diagonal, eig, keep >0
//
// MI355X mapping:
//   zero/pos   one elementwise kernel;
//   box        ONE workgroup of 1024 lanes runs the whole Newton loop (a global
//              reduction per step is a workgroup reduction; bounds stream from L2);
//   SOC        cones are few-and-huge on the headline config (17 cones of ~7e4
//              rows) and many-and-tiny elsewhere, so "one workgroup per cone" is
//              wrong both ways: cones with q <= 16 get one lane each (sequential,
//              reference summation order); everything else is cut into 2048-row
//              tiles -> per-tile partial, per-cone finalize, per-tile apply;
//   PSD        one workgroup per cone: parallel cyclic two-sided Jacobi on the
//              k x k matrix held in LDS (k=50 -> 41 KB of 160 KB), eigenvector
//              accumulation in LDS, then X+ = V diag(max(l,0)) V' on the fp64
//              matrix cores (v_mfma_f64_16x16x4_f64) -- replaces LAPACK
//              dsyevr + dsyrk (cones.c:1028,1052).
#include "cones.h"
#include "cones_exp_pow.h"
#include <algorithm>
#include <type_traits>
#include "psd_lds_step.h" // RotCS, jacobi_cs, psd_update_pass, the look-ahead of the pipelined step

namespace scsamd {

constexpr int SOC_TINY_MAX = 16;
constexpr int SOC_TILE = 2048;
constexpr int BOX_THREADS = 1024;
constexpr int BOX_MAX_ITERS = 25;     // BOX_CONE_MAX_ITERS, cones.c:21
constexpr double MAX_BOX_VAL = 1e15;  // cones.c:54
constexpr int PSD_THREADS = 512;
constexpr int PSD_LDS_KMAX = 92;      // 2*92*93*8 B + 12 KB header = 149 KB < 160 KB
constexpr int PSD_MAX_SWEEPS = 30;
constexpr int PSD_WARM_KMAX = 72;     // warm start keeps a third K2 x ld matrix in LDS: 3*72*73*8 B + header < 160 KB
constexpr int PSD_WARM_RESET = 64;    // cold restart period: bounds the orthogonality drift of the carried basis
constexpr int PSD_MAX_PAIRS = 512;    // supports k <= 1024
// -DSCSAMD_PSD_CLOCKS (scripts/build_variant.sh): cone 0 of every launch prints its phase clocks (clock64: shader clock) and step counts
// -- the measurement behind profiles/r5_psd_pipelined_step.md; compiled out of every shipped library
#ifdef SCSAMD_PSD_CLOCKS
#define PSD_CLK(v) const long long v = clock64()
#define PSD_COUNT(x) ++(x)
#else
#define PSD_CLK(v)
#define PSD_COUNT(x)
#endif
constexpr int PSD_TBL = 256;          // second rotation-table buffer of the pipelined step (LDS path: <= 46 pairs)
#ifdef SCSAMD_PSD_LA_ALONE
// measurement variant (round 6): the look-ahead wave (wave 7) ALONE on its SIMD -- wave 3, which shares SIMD 3 with it (waves go to the
// SIMDs round robin), only keeps the barriers company; six update waves, four rows of V per lane
constexpr int PSD_PIPE_THREADS = PSD_THREADS - 128;
constexpr int PSD_PIPE_VR = 4;
#else
constexpr int PSD_PIPE_THREADS = PSD_THREADS - 64; // update lanes of the pipelined step (the last wave looks ahead)
constexpr int PSD_PIPE_VR = 3;
#endif
constexpr int PSD_K_LIMIT = 2 * PSD_MAX_PAIRS;
constexpr size_t PSD_LDS_HEADER = PSD_MAX_PAIRS * (2 * sizeof(real) + 2 * sizeof(int)) + 12 * sizeof(real); // tables, red[8], 4 words of flags (fp32: red + 8 .. red + 11)

// ----------------------------------------------------------------------------
// Moreau pre / post (cones.c:1567-1593)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_moreau_pre(real *x, real *s, const real *__restrict__ ry,
                                                             int m) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    const real xi = x[i];
    s[i] = xi;
    x[i] = ry ? xi * (-ry[i]) : -xi;
  }
}
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_moreau_post(real *x, const real *__restrict__ s,
                                                              const real *__restrict__ ry, int m) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
    x[i] = ry ? x[i] / ry[i] + s[i] : x[i] + s[i];
}

// zero cone -> 0, nonnegative orthant -> max(x, 0)   (cones.c:1349-1359)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_zero_pos(real *x, int z, int l) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < z + l; i += gridDim.x * blockDim.x) {
    if (i < z) x[i] = 0;
    else {
      const real v = x[i];
      x[i] = v > (real)0 ? v : (real)0;
    }
  }
}

// ----------------------------------------------------------------------------
// box cone: whole Newton loop in one workgroup
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(BOX_THREADS) void k_box(real *tx, const real *__restrict__ bl,
                                                     const real *__restrict__ bu, int bsize,
                                                     real *t_warm, const real *r_box) {
  __shared__ real red[BOX_THREADS / SCSAMD_WAVE];
  const int tid = threadIdx.x;
  if (bsize == 1) {
    if (tid == 0) tx[0] = tx[0] > (real)0 ? tx[0] : (real)0;
    return;
  }
  real *x = tx + 1;
  const real *rho = r_box ? r_box + 1 : nullptr;
  const real rho_t = r_box ? (real)1 / r_box[0] : (real)1;
  const real t0 = tx[0];
  real t = t_warm[0];
  for (int it = 0; it < BOX_MAX_ITERS; ++it) {
    const real t_prev = t;
    real gt = 0, ht = 0;
    for (int j = tid; j < bsize - 1; j += BOX_THREADS) {
      const real r = rho ? (real)1 / rho[j] : (real)1;
      const real xj = x[j], u = bu[j], lo = bl[j];
      if (xj > t * u) {
        gt += r * (t * u - xj) * u;
        ht += r * u * u;
      } else if (xj < t * lo) {
        gt += r * (t * lo - xj) * lo;
        ht += r * lo * lo;
      }
    }
    gt = block_sum(gt, red);
    ht = block_sum(ht, red);
    gt += rho_t * (t - t0);
    ht += rho_t;
    const real hm = ht > (real)1e-8 ? ht : (real)1e-8;
    t = t - gt / hm;
    t = t > (real)0 ? t : (real)0;
    const real hm6 = ht > (real)1e-6 ? ht : (real)1e-6;
    const real tm = t > (real)1 ? t : (real)1;
    if (absval(gt / hm6) < (real)1e-12 * tm || absval(t - t_prev) < (real)1e-11 * tm) break;
  }
  for (int j = tid; j < bsize - 1; j += BOX_THREADS) {
    const real xj = x[j], u = bu[j], lo = bl[j];
    if (xj > t * u) x[j] = t * u;
    else if (xj < t * lo) x[j] = t * lo;
  }
  if (tid == 0) {
    tx[0] = t;
    t_warm[0] = t;
  }
}

// ----------------------------------------------------------------------------
// box cone, large bsize: the same Newton iteration spread over the chip.  One launch per Newton step
// (k_box_step) in the device-controlled pattern of the PCG: step `it` first finishes step it-1 -- every
// workgroup re-reduces the previous step's per-workgroup partials of (g, h) in the same fixed order, so all
// of them hold the same new t; workgroup 0 publishes it -- and, unless the stop tests of
// src/cones.c:1231-1232 fired, forms its slice's partials at the new t.  Steps enqueued past convergence
// return at once.  k_box_apply clamps x with the final t.  A 1e6-row box costs ~4 passes over 24 MB on the
// whole chip instead of 25 on one CU.
// ----------------------------------------------------------------------------
struct BoxCtl {
  real t[2];  // Newton iterate published by step it, slot it & 1
  real t_cur; // latest published iterate (what k_box_apply uses)
  int done;
};
constexpr int BOX_MULTI_MIN = 16384; // rows from which the multi-workgroup path is used
constexpr int BOX_MULTI_GRID = 256;

__global__ __launch_bounds__(SCSAMD_BLOCK) void k_box_step(const real *__restrict__ tx, const real *__restrict__ bl,
                                                            const real *__restrict__ bu, int bsize,
                                                            const real *t_warm, const real *r_box, BoxCtl *ctl,
                                                            real *part, int it) {
  __shared__ real red[SCSAMD_BLOCK / SCSAMD_WAVE];
  const int tid = threadIdx.x, nwg = gridDim.x;
  const real *x = tx + 1;
  const real *rho = r_box ? r_box + 1 : nullptr;
  real t;
  if (it == 0) {
    t = t_warm[0];
    if (blockIdx.x == 0 && tid == 0) {
      ctl->t[0] = t;
      ctl->t_cur = t;
      ctl->done = 0;
    }
  } else {
    if (ctl->done) return;
    const real t_prev = ctl->t[(it - 1) & 1];
    const real *pg = part + (size_t)((it - 1) & 1) * 2 * BOX_MULTI_GRID, *ph = pg + BOX_MULTI_GRID;
    real gt = reduce_partials_sum(pg, nwg, red);
    real ht = reduce_partials_sum(ph, nwg, red);
    const real rho_t = r_box ? (real)1 / r_box[0] : (real)1;
    gt += rho_t * (t_prev - tx[0]);
    ht += rho_t;
    const real hm = ht > (real)1e-8 ? ht : (real)1e-8;
    t = t_prev - gt / hm;
    t = t > (real)0 ? t : (real)0;
    const real hm6 = ht > (real)1e-6 ? ht : (real)1e-6;
    const real tm = t > (real)1 ? t : (real)1;
    const bool stop = absval(gt / hm6) < (real)1e-12 * tm || absval(t - t_prev) < (real)1e-11 * tm || it >= BOX_MAX_ITERS;
    if (blockIdx.x == 0 && tid == 0) {
      ctl->t[it & 1] = t;
      ctl->t_cur = t;
      if (stop) ctl->done = 1;
    }
    if (stop) return;
  }
  real gt = 0, ht = 0;
  for (int j = blockIdx.x * SCSAMD_BLOCK + tid; j < bsize - 1; j += nwg * SCSAMD_BLOCK) {
    const real r = rho ? (real)1 / rho[j] : (real)1;
    const real xj = x[j], u = bu[j], lo = bl[j];
    if (xj > t * u) {
      gt += r * (t * u - xj) * u;
      ht += r * u * u;
    } else if (xj < t * lo) {
      gt += r * (t * lo - xj) * lo;
      ht += r * lo * lo;
    }
  }
  gt = block_sum(gt, red);
  ht = block_sum(ht, red);
  if (tid == 0) {
    real *pg = part + (size_t)(it & 1) * 2 * BOX_MULTI_GRID;
    pg[blockIdx.x] = gt;
    pg[BOX_MULTI_GRID + blockIdx.x] = ht;
  }
}

__global__ __launch_bounds__(SCSAMD_BLOCK) void k_box_apply(real *tx, const real *__restrict__ bl,
                                                             const real *__restrict__ bu, int bsize, real *t_warm,
                                                             const BoxCtl *ctl) {
  const real t = ctl->t_cur;
  real *x = tx + 1;
  for (int j = blockIdx.x * SCSAMD_BLOCK + threadIdx.x; j < bsize - 1; j += gridDim.x * SCSAMD_BLOCK) {
    const real xj = x[j], u = bu[j], lo = bl[j];
    if (xj > t * u) x[j] = t * u;
    else if (xj < t * lo) x[j] = t * lo;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    tx[0] = t;
    t_warm[0] = t;
  }
}

// ----------------------------------------------------------------------------
// second-order cones
// ----------------------------------------------------------------------------
__device__ __forceinline__ void soc_decide(real v1, real s, real &head, real &mult) {
  if (s <= v1) { // inside (cones.c:1271)
    head = v1;
    mult = 1;
  } else if (s <= -v1) { // polar (cones.c:1273)
    head = 0;
    mult = 0;
  } else {
    const real alpha = (s + v1) / (real)2;
    head = alpha;
    mult = alpha / s;
  }
}

// one lane per tiny cone; same case analysis and summation order as proj_soc
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_soc_tiny(real *x, const int *__restrict__ off,
                                                           const int *__restrict__ len, int ncones) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncones; c += gridDim.x * blockDim.x) {
    real *xc = x + off[c];
    const int q = len[c];
    if (q <= 0) continue;
    if (q == 1) {
      xc[0] = xc[0] > (real)0 ? xc[0] : (real)0;
      continue;
    }
    const real v1 = xc[0];
    real s;
    if (q == 2) s = absval(xc[1]);
    else {
      real ss = 0;
      for (int j = 1; j < q; ++j) ss += xc[j] * xc[j];
      s = sqrt(ss);
    }
    real head, mult;
    soc_decide(v1, s, head, mult);
    if (mult == (real)1 && head == v1) continue;
    xc[0] = head;
    if (mult == (real)0)
      for (int j = 1; j < q; ++j) xc[j] = 0;
    else
      for (int j = 1; j < q; ++j) xc[j] *= mult;
  }
}

// pass 1: per-tile sum of squares of tail entries
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_soc_tile_partial(const real *__restrict__ x,
                                                                   const int *__restrict__ tile_off,
                                                                   const int *__restrict__ tile_len,
                                                                   const int *__restrict__ tile_cone,
                                                                   const int *__restrict__ big_off,
                                                                   real *part, int ntiles) {
  __shared__ real red[4];
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int o = tile_off[t], n = tile_len[t];
    const int head = big_off[tile_cone[t]];
    real ss = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const real v = x[o + j];
      if (o + j != head) ss += v * v;
    }
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) part[t] = ss;
  }
}
// pass 2: one lane per cone
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_soc_finalize(const real *__restrict__ x,
                                                               const int *__restrict__ big_off,
                                                               const int *__restrict__ big_tile0,
                                                               const real *__restrict__ part, real *coef,
                                                               int ncones) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncones; c += gridDim.x * blockDim.x) {
    real ss = 0;
    for (int t = big_tile0[c]; t < big_tile0[c + 1]; ++t) ss += part[t];
    const real s = sqrt(ss), v1 = x[big_off[c]];
    real head, mult;
    soc_decide(v1, s, head, mult);
    coef[2 * c] = head;
    coef[2 * c + 1] = mult;
  }
}
// pass 3: apply
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_soc_tile_apply(real *x, const int *__restrict__ tile_off,
                                                                 const int *__restrict__ tile_len,
                                                                 const int *__restrict__ tile_cone,
                                                                 const int *__restrict__ big_off,
                                                                 const real *__restrict__ coef,
                                                                 int ntiles) {
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int o = tile_off[t], n = tile_len[t], c = tile_cone[t];
    const int head = big_off[c];
    const real hv = coef[2 * c], mult = coef[2 * c + 1];
    if (mult == (real)1) continue; // inside the cone: untouched (head == v1)
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      if (o + j == head) x[o + j] = hv;
      else x[o + j] = mult == (real)0 ? (real)0 : x[o + j] * mult;
    }
  }
}

// ----------------------------------------------------------------------------
// PSD cone: parallel cyclic Jacobi eigensolver, one workgroup per cone
// ----------------------------------------------------------------------------
#ifndef SFLOAT
typedef double f64x4 __attribute__((ext_vector_type(4)));
#endif

// packed lower triangle, column major: column j starts at j*k - j*(j-1)/2
__device__ __forceinline__ int packed_index(int i, int j, int k) { // i >= j
  return j * k - (j * (j - 1)) / 2 + (i - j);
}

// C(K2 x K2, leading dim ld, in LDS) = op(L) * R with op(L) = L or L', all K2 x K2 in LDS.
// fp64: v_mfma_f64_16x16x4_f64 tiles (lane l supplies L-operand (row l&15, k l>>4) and R-operand
// (k l>>4, col l&15); D element (row (l>>4) + 4 reg, col l&15)); fp32 build: plain loops.
template <bool TRANS_L>
__device__ __forceinline__ void psd_lds_matmul(real *C, const real *L, const real *R, int K2, int ld, int tid) {
#ifndef SFLOAT
  const int wave = tid >> 6, lane = tid & 63, nw = PSD_THREADS >> 6;
  const int T = (K2 + 15) >> 4, ksteps = (K2 + 3) >> 2;
  const int li = lane & 15, lk = lane >> 4;
  for (int t = wave; t < T * T; t += nw) {
    const int ti = t / T, tj = t % T;
    f64x4 acc = {0, 0, 0, 0};
    const int ra = ti * 16 + li, cb = tj * 16 + li;
    for (int ks = 0; ks < ksteps; ++ks) {
      const int kc = ks * 4 + lk;
      const bool ok = kc < K2;
      const double av = (ok && ra < K2) ? (TRANS_L ? L[kc * ld + ra] : L[ra * ld + kc]) : 0.0;
      const double bv = (ok && cb < K2) ? R[kc * ld + cb] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ti * 16 + lk + 4 * r, j = tj * 16 + li;
      if (i < K2 && j < K2) C[i * ld + j] = acc[r];
    }
  }
#else
  for (int e = tid; e < K2 * K2; e += PSD_THREADS) {
    const int i = e / K2, j = e % K2;
    real acc = 0;
    for (int kc = 0; kc < K2; ++kc) acc += (TRANS_L ? L[kc * ld + i] : L[i * ld + kc]) * R[kc * ld + j];
    C[i * ld + j] = acc;
  }
#endif
}

// entry (i, j) of the full symmetric matrix behind a packed block, diagonal * sqrt(2) (cones.c:1018-1025); for a
// complex Hermitian block of order nn (k = 2 nn) the entry of its real symmetric embedding [[A, -B], [B, A]]
__device__ __forceinline__ real psd_unpack_entry(const real *X, int k, bool cplx, int nn, int i, int j) {
  const real sqrt2 = sqrt((real)2);
  real v = 0;
  if (i < k && j < k) {
    if (!cplx) {
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      v = X[packed_index(hi, lo, k)];
      if (i == j) v *= sqrt2;
    } else {
      // column c of the packed Hermitian lower triangle starts at c (2 nn - c): real diagonal,
      // then (re, im) pairs of rows c+1.. (cones.c:1095-1103)
      const int bi = i / nn, bj = j / nn, ii = i % nn, jj = j % nn;
      const int r = ii > jj ? ii : jj, c = ii > jj ? jj : ii;
      if (bi == bj) { // A block: real part, symmetric
        v = r == c ? X[c * (2 * nn - c)] * sqrt2 : X[c * (2 * nn - c) + 1 + 2 * (r - c - 1)];
      } else if (r != c) { // +-B block: imaginary part, antisymmetric
        real im = X[c * (2 * nn - c) + 2 + 2 * (r - c - 1)]; // Im H[r][c], r > c
        if (ii < jj) im = -im;                                // B[ii][jj] = -B[jj][ii]
        v = (bi == 1 && bj == 0) ? im : -im;                  // M[i+nn][j] = B, M[i][j+nn] = -B
      }
    }
  }
  return v;
}

// vprev (nullable): per cone a K2m x ldm eigenbasis carried from the previous projection.  With
// warm != 0 the iteration starts from A' = Vp' A Vp (nearly diagonal when consecutive ADMM
// iterates are close) and V = Vp, so it needs 1-2 sweeps instead of ~8; the basis is written back
// whenever vprev is given.  The host restarts cold every PSD_WARM_RESET calls.
// lane i <- lane i + 1 (UP) / lane i - 1 (!UP) of a wave, by DPP (v_mov_b32_dpp wave_shl:1 / wave_shr:1): a register move, NOT an LDS
// instruction (ds_bpermute would queue behind the update waves' LDS traffic, which is what the look-ahead must avoid).  All 64 lanes
// must be active at the call.  The first / last lane keeps its own value.
template <bool UP>
__device__ __forceinline__ int lane_shift1(int v) {
  return UP ? __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false) : __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}
template <bool UP>
__device__ __forceinline__ PsdRot lane_shift1(const PsdRot &r) {
  PsdRot o;
  o.x = lane_shift1<UP>(r.x);
  o.y = lane_shift1<UP>(r.y);
  union {
    real f;
    int w[sizeof(real) / 4];
  } a, b;
  a.f = r.c;
#pragma unroll
  for (unsigned j = 0; j < sizeof(real) / 4; ++j) b.w[j] = lane_shift1<UP>(a.w[j]);
  o.c = b.f;
  a.f = r.s;
#pragma unroll
  for (unsigned j = 0; j < sizeof(real) / 4; ++j) b.w[j] = lane_shift1<UP>(a.w[j]);
  o.s = b.f;
  a.f = r.t;
#pragma unroll
  for (unsigned j = 0; j < sizeof(real) / 4; ++j) b.w[j] = lane_shift1<UP>(a.w[j]);
  o.t = b.f;
  return o;
}

// PIPE: the pipelined step (round 5) for launches whose largest block leaves room for a second copy of A (order <= PSD_WARM_KMAX);
// its own instantiation, so that the five-blocks-per-lane update of orders up to 92 does not set this one's register budget
// PIPE = 0: two-phase step; 1: round 5's look-ahead (the next rotations re-derived from the matrix before the step); 2: round 6's signal
// form (the update lanes own the blocks the next rotations need FIRST and raise a counter; the rotation wave reads the updated entries)
template <int PIPE>
__global__ __launch_bounds__(PSD_THREADS) void k_psd_jacobi(real *x, const int *__restrict__ psd_off,
                                                            const int *__restrict__ psd_k, real *scratch,
                                                            int kmax, int lds_kmax, int *status, real *vprev,
                                                            int warm) {
  // all scratch lives in the dynamic region (16-byte aligned base, guide G17):
  // [rot_cs (c,s pairs) | rot_pq (p,q pairs) | red | step flags | A | V]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RotCS *rot_cs = reinterpret_cast<RotCS *>(smem_raw);
  int2 *rot_pq = reinterpret_cast<int2 *>(rot_cs + PSD_MAX_PAIRS);
  real *red = reinterpret_cast<real *>(rot_pq + PSD_MAX_PAIRS);
  // [2]: does step (parity) rotate at all?  NOT volatile: LLVM leaves volatile accesses in the generic address space, i.e. FLAT
  // loads / stores (flat_load_dword sc0 sc1 + s_waitcnt vmcnt(0)) with vector-memory latency on the critical path of EVERY step --
  // rounds 2-4 paid that in the two-phase step too.  Every write and its reads are separated by a workgroup barrier.
  int *rot_any = reinterpret_cast<int *>(red + 8);
  real *lds_mat = red + 12;
  const int cone = blockIdx.x, tid = threadIdx.x;
  // psd_k > 0: real symmetric block of that order (packed lower triangle).
  // psd_k < 0: complex Hermitian block of order nn = -psd_k/2 (src/cones.c:1072-1155), handled
  //            through its real symmetric embedding M = [[A, -B], [B, A]] of order 2 nn
  //            (H = A + iB): M's eigenvalues are H's, doubled, and the PSD part of M is the
  //            embedding of the PSD part of H, so the same Jacobi iteration serves both.
  const int kraw = psd_k[cone];
  const bool cplx = kraw < 0;
  const int k = cplx ? -kraw : kraw;
  const int nn = k / 2; // complex order (cplx only)
  real *X = x + psd_off[cone];
  if (k <= 0) return;
  if (k == 1 || (cplx && nn == 1)) {
    if (tid == 0) X[0] = X[0] > (real)0 ? X[0] : (real)0;
    return;
  }
  // pad to an even order K2 with a zero row/column: the odd index pairs with the dummy
  // (identity rotation), so every step is npairs full 2x2-block updates
  const int K2 = (k + 1) & ~1;
  const int npairs = K2 / 2;
  const int ld = K2 | 1; // odd leading dimension: conflict-free row and column walks
  const int K2m = (kmax + 1) & ~1, ldm = K2m | 1;
  const bool use_lds = k <= lds_kmax; // per block: small blocks stay in LDS next to a large one in the same program
  if (!use_lds) return; // larger blocks: psd_big.h (chip-wide steps)
  real *A = lds_mat;
  real *V = A + (size_t)K2 * ld;
  real *Acur = A; // the pipelined step ping-pongs between A and a second copy behind V
  // element (r, c): in LDS row-major with an odd leading dimension (lanes that walk rows hit distinct banks); in the
  // global-memory scratch of large blocks column-major, so the same lanes touch consecutive addresses (the 2x2-block
  // pass and the eigenvector update walk rows -- row-major there cost a 128-byte line per 8-byte access)
  auto MI = [&](int r, int c) { return use_lds ? r * ld + c : c * ld + r; };
  const real sqrt2 = sqrt((real)2);
  PSD_CLK(clk0);
#ifdef SCSAMD_PSD_CLOCKS
  int n_steps = 0, n_rot_steps = 0;
  long long clk_work = 0, clk_wait = 0; // per wave: inside a step before its barrier / waiting at the barrier (pipelined step)
#endif
  // unpack: full symmetric, diagonal * sqrt(2)  (cones.c:1018-1025)
  for (int e = tid; e < K2 * K2; e += PSD_THREADS) {
    const int i = e % K2, j = e / K2;
    const real v = psd_unpack_entry(X, k, cplx, nn, i, j);
    A[MI(i, j)] = v;
    V[MI(i, j)] = i == j ? (real)1 : (real)0;
  }
  __syncthreads();
  real *Vg = vprev ? vprev + (size_t)cone * K2m * ldm : nullptr;
  if (Vg && warm) {
    // V <- Vp;  T <- A Vp;  A <- Vp' T, symmetrised (the rotations assume A == A' exactly).  T sits behind V in LDS while
    // three matrices fit (order <= PSD_WARM_KMAX); above that (73..92) it goes through this cone's slice of an HBM scratch --
    // written and read back by the same workgroup, so it stays in this CU's L1 / the XCD's L2 (round 4: orders 73..92 ran
    // ~8 cold sweeps per projection before, 64 us per block at order 92 against 3.4 us at order 64)
    real *Tm = scratch ? scratch + (size_t)cone * K2m * ldm : V + (size_t)K2 * ld;
    for (int e = tid; e < K2 * K2; e += PSD_THREADS) {
      const int i = e / K2, j = e % K2;
      V[i * ld + j] = Vg[i * ld + j];
    }
    __syncthreads();
    psd_lds_matmul<false>(Tm, A, V, K2, ld, tid);
    __syncthreads();
    psd_lds_matmul<true>(A, V, Tm, K2, ld, tid);
    __syncthreads();
    for (int e = tid; e < K2 * K2; e += PSD_THREADS) {
      const int i = e / K2, j = e % K2;
      if (i > j) {
        const real v = (real)0.5 * (A[i * ld + j] + A[j * ld + i]);
        A[i * ld + j] = v;
        A[j * ld + i] = v;
      }
    }
    __syncthreads();
  }
  PSD_CLK(clk1);
  real fro = 0;
  for (int e = tid; e < K2 * K2; e += PSD_THREADS) {
    const real v = A[MI(e % K2, e / K2)];
    fro += v * v;
  }
  fro = sqrt(block_sum(fro, red));
  const real eps = sizeof(real) == 8 ? (real)1e-15 : (real)1e-7;
  PSD_CLK(clk2);
  int sweep = 0;
  if (fro > (real)0) {
    // an off-diagonal entry at or below the convergence threshold is left alone; a step
    // whose pairs are all below it skips the update pass (and its barrier) altogether, so
    // the last, verifying sweep costs only the pair scan
    // fp32: entries cannot be driven below the rounding noise of the rotations themselves (~eps_mach |A|);
    // a threshold under it would run all PSD_MAX_SWEEPS sweeps on every call
    const real thr = sizeof(real) == 8 ? eps * fro / (real)k : fmaxf(eps * fro / (real)k, (real)2.4e-7 * fro);
    if (tid < 2) rot_any[tid] = 0;
    __syncthreads();
    // pipelined step: needs a second copy of A (the warm start's T region behind V, dead by now) -> orders up to PSD_WARM_KMAX of a
    // launch whose largest LDS block is that small (the host sizes the LDS for three matrices then); K2 = 2 has a single step
    const bool pipelined = PIPE != 0 && K2 <= PSD_WARM_KMAX && K2 >= 4;
    const int wave = tid >> 6, lane = tid & 63;
    const bool la = wave == PSD_THREADS / 64 - 1; // the look-ahead wave
    real *A2 = V + (size_t)K2 * ld;
    // (round 4) a sweep that would rotate nothing is not run: one pass over the off-diagonal entries instead of K2 - 1 steps of pair
    // scans.  The warm-started iteration of consecutive ADMM iterates ends with exactly such a verifying sweep; skipping it leaves the
    // same A and V (a sweep without rotations changes nothing).
    auto nothing_to_rotate = [&]() {
      real m = 0;
      for (int e = tid; e < K2 * K2; e += PSD_THREADS) {
        const int i = e / K2, j = e % K2;
        if (i < j && j < k) {
          const real v = absval(Acur[MI(i, j)]);
          m = v > m ? v : m;
        }
      }
      return block_max(m, red) <= thr;
    };
    // ---- pipelined step (round 5; psd_lds_step.h): the last wave forms step s+1's rotations from step s's tables and the matrix as it
    // stands BEFORE step s, while the other seven waves apply step s from A[cur] into the other copy -- ONE barrier per step instead
    // of two, and the rotation chain (two rsqrt sequences on <= 36 lanes) off the critical path.  NB = 2x2 blocks per update lane.
    auto sweeps_pipelined = [&](auto nbc) {
      constexpr int NB = decltype(nbc)::value;
      PsdItems<NB, PSD_PIPE_VR> items;
#ifdef SCSAMD_PSD_LA_ALONE
      const int utid = wave < 3 ? tid : (wave == 3 ? PSD_PIPE_THREADS : tid - 64); // wave 3 idle: an index no item belongs to
#else
      const int utid = tid;
#endif
      psd_items_init<NB, PSD_PIPE_VR>(items, utid, PSD_PIPE_THREADS, npairs, K2); // the lane's blocks and row pairs: once, not per step
      if (la) __builtin_amdgcn_s_setprio(3); // the look-ahead wave's chain is the longer one: it issues first on its SIMD
      for (; sweep < PSD_MAX_SWEEPS; ++sweep) {
        if (nothing_to_rotate()) break;
        real offmax = 0;
        int pos_a = lane, pos_b = K2 - 1 - lane; // step 0 (lane = pair index in the look-ahead wave)
        PsdRot mine{0, 1, (real)1, (real)0, (real)0}; // look-ahead lane i: pair i of the step being applied, as it wrote it to the tables
        PsdLaPlan plan;                                // ... and what it will read / multiply for its pair of the NEXT step (psd_la_prepare)
        int la_any = 0;                                // look-ahead wave: does that step rotate at all (its own vote, no LDS read)
        // the records of the pairs that hold this lane's next players are its neighbours' registers (DPP; all 64 lanes take part);
        // formed right after `mine`, i.e. BEFORE the barrier that opens the phase in which the plan's reads are issued
        auto la_prepare = [&]() {
          const PsdRot up = lane_shift1<true>(mine), dn = lane_shift1<false>(mine);
          if (lane < npairs) {
            // pair i of the next step takes its first player from pair i + 1 (the last pair: from itself) and its second from pair
            // i - 1.  Pair 0 is the exception (player 0 stays, the other comes from pair 1): it names its players the other way round,
            // so that for EVERY lane the first record is the upper neighbour's and the second the lower one's -- lane 0's own, because a
            // shift down leaves the first lane its own value.  (Pairs are unordered; one select less per record field.)
            const PsdRot rec_p = lane == npairs - 1 ? mine : up;
            const int p = lane == 0 ? pos_b : pos_a, q = lane == 0 ? pos_a : pos_b;
            psd_la_prepare(plan, rec_p, dn, p, q, ld);
          }
        };
        if (la) { // prologue: step 0 from the matrix as it stands
          bool rot = false;
          if (lane < npairs) {
            rot = psd_first_rotation(Acur, pos_a, pos_b, ld, k, thr, offmax, mine);
            psd_pair_advance(lane, K2, pos_a, pos_b);
            rot_pq[lane] = make_int2(mine.x, mine.y);
            rot_cs[lane] = RotCS{mine.c, mine.s};
          }
          la_any = __any(rot ? 1 : 0);
          if (lane == 0) rot_any[0] = la_any;
          if (K2 - 1 > 1) la_prepare();
        }
        __syncthreads();
        for (int step = 0; step < K2 - 1; ++step) {
          const int par = step & 1;
          const int2 *tq = rot_pq + par * PSD_TBL;
          const RotCS *tc = rot_cs + par * PSD_TBL;
          real *Anext = Acur == A ? A2 : A;
          bool rotates;
          PSD_CLK(clk_a);
          if (la) {
            rotates = la_any != 0;
            if (step + 1 < K2 - 1) {
              bool rot = false;
              if (lane < npairs) {
                rot = psd_la_finish(Acur, plan, k, thr, offmax, mine); // its eight reads open the phase's LDS queue
                psd_pair_advance(lane, K2, pos_a, pos_b);
                rot_pq[(par ^ 1) * PSD_TBL + lane] = make_int2(mine.x, mine.y);
                rot_cs[(par ^ 1) * PSD_TBL + lane] = RotCS{mine.c, mine.s};
              }
              la_any = __any(rot ? 1 : 0);
              if (lane == 0) rot_any[par ^ 1] = la_any;
              if (step + 2 < K2 - 1) la_prepare(); // for the phase after the barrier
            }
          } else {
            rotates = psd_update_pass<NB, PSD_PIPE_VR>(Acur, Anext, V, tq, tc, items, ld, rot_any + par) != 0; // the flag is read with the tables (uniform)
          }
#ifdef SCSAMD_PSD_CLOCKS
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const long long clk_b = clock64();
#endif
          __syncthreads();
#ifdef SCSAMD_PSD_CLOCKS
          clk_work += clk_b - clk_a;
          clk_wait += clock64() - clk_b;
#endif
          PSD_COUNT(n_steps);
          if (rotates) {
            Acur = Anext;
            PSD_COUNT(n_rot_steps);
          }
        }
        offmax = block_max(offmax, red);
        if (offmax <= thr) break;
      }
      if (la) __builtin_amdgcn_s_setprio(0);
    };
    // ---- signal form (round 6; psd_lds_step.h): as above, ONE barrier per step and a ping-pong copy of A, but the rotation wave does not
    // re-derive its entries: the update lanes own the 2 npairs priority blocks as their first items, store them and raise `sig`; the
    // rotation wave waits for `sig` and reads (a_pq, a_pp, a_qq) of its next pair from the updated copy -- psd_first_rotation's plain
    // formula on the stored values: the iteration is bit-identical to the two-phase one.
    auto sweeps_signal = [&](auto nbc) {
      constexpr int NB = decltype(nbc)::value;
      PsdItems<NB, 3> items;
      psd_items_init_priority<NB, 3>(items, tid, PSD_PIPE_THREADS, npairs, K2);
      int *sig = rot_any + 2; // (header: two flag words, then this counter; `red + 12` starts the matrices)
      const int npri = psd_priority_count(npairs);
      const int nsig = (npri < PSD_PIPE_THREADS ? npri + 63 : PSD_PIPE_THREADS + 63) >> 6; // waves whose first items are priority blocks
      int *my_signal = (!la && wave < nsig && lane == 0) ? sig : nullptr;
      if (la) __builtin_amdgcn_s_setprio(3);
      for (; sweep < PSD_MAX_SWEEPS; ++sweep) {
        if (nothing_to_rotate()) break;
        real offmax = 0;
        int pos_a = lane, pos_b = K2 - 1 - lane;
        int la_any = 0, expected = 0;
        if (tid == 0) *sig = 0; // (the barriers of nothing_to_rotate separate this from the last sweep's reads)
        if (la) { // prologue: step 0 from the matrix as it stands
          bool rot = false;
          if (lane < npairs) {
            PsdRot r;
            rot = psd_rotation_now(Acur, pos_a, pos_b, ld, k, thr, offmax, r);
            psd_pair_advance(lane, K2, pos_a, pos_b);
            rot_pq[lane] = make_int2(r.x, r.y);
            rot_cs[lane] = RotCS{r.c, r.s};
          }
          la_any = __any(rot ? 1 : 0);
          if (lane == 0) rot_any[0] = la_any;
        }
        __syncthreads();
        for (int step = 0; step < K2 - 1; ++step) {
          const int par = step & 1;
          const int2 *tq = rot_pq + par * PSD_TBL;
          const RotCS *tc = rot_cs + par * PSD_TBL;
          real *Anext = Acur == A ? A2 : A;
          bool rotates;
          PSD_CLK(clk_a);
          if (la) {
            rotates = la_any != 0;
            if (step + 1 < K2 - 1) {
              const real *Aread = Acur; // a step that rotates nothing leaves the matrix where it is
              if (rotates) {
                expected += nsig;
                while (__hip_atomic_load(sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < expected) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                Aread = Anext;
              }
              bool rot = false;
              if (lane < npairs) {
                PsdRot r;
                rot = psd_rotation_now(Aread, pos_a, pos_b, ld, k, thr, offmax, r);
                psd_pair_advance(lane, K2, pos_a, pos_b);
                rot_pq[(par ^ 1) * PSD_TBL + lane] = make_int2(r.x, r.y);
                rot_cs[(par ^ 1) * PSD_TBL + lane] = RotCS{r.c, r.s};
              }
              la_any = __any(rot ? 1 : 0);
              if (lane == 0) rot_any[par ^ 1] = la_any;
            }
          } else {
            rotates = psd_update_pass<NB, 3, true>(Acur, Anext, V, tq, tc, items, ld, rot_any + par, my_signal) != 0;
          }
#ifdef SCSAMD_PSD_CLOCKS
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const long long clk_b = clock64();
#endif
          __syncthreads();
#ifdef SCSAMD_PSD_CLOCKS
          clk_work += clk_b - clk_a;
          clk_wait += clock64() - clk_b;
#endif
          PSD_COUNT(n_steps);
          if (rotates) {
            Acur = Anext;
            PSD_COUNT(n_rot_steps);
          }
        }
        offmax = block_max(offmax, red);
        if (offmax <= thr) break;
      }
      if (la) __builtin_amdgcn_s_setprio(0);
    };
    // ---- two-phase step (rounds 2-4): rotation parameters on the first npairs lanes, barrier, in-place update, barrier.  Orders
    // 73..92 (no room for a second copy of A) and K2 = 2.
    // round-robin pairing: player 0 fixed, the others rotate -- pair i of step s is (0 or 1 + (i - 1 + s) mod (K2 - 1),
    // 1 + (K2 - 2 - i + s) mod (K2 - 1)); the two positions advance by one per step (no run-time modulus on the step's critical path)
    auto sweeps_two_phase = [&](auto nbc) {
      constexpr int NB = decltype(nbc)::value;
      PsdItems<NB, 3> items;
      psd_items_init<NB, 3>(items, tid, PSD_THREADS, npairs, K2);
      for (; sweep < PSD_MAX_SWEEPS; ++sweep) {
        if (nothing_to_rotate()) break;
        real offmax = 0;
        int pos_a = tid, pos_b = K2 - 1 - tid; // step 0 (for tid < npairs: i - 1 < K2 - 1 and K2 - 2 - i >= 0)
        for (int step = 0; step < K2 - 1; ++step) {
          const int par = step & 1;
          if (tid < npairs) {
            PsdRot r;
            // an off-diagonal entry at or below the threshold is left alone (identity); t = sgn(d) b / (|d| + sqrt(d^2 + b^2))
            if (psd_first_rotation(A, pos_a, pos_b, ld, k, thr, offmax, r)) rot_any[par] = 1;
            psd_pair_advance(tid, K2, pos_a, pos_b);
            rot_pq[tid] = make_int2(r.x, r.y);
            rot_cs[tid] = RotCS{r.c, r.s};
            if (tid == 0) rot_any[par ^ 1] = 0; // nobody reads the other parity before the next step's barrier
          }
          __syncthreads();
          PSD_COUNT(n_steps);
          if (!rot_any[par]) continue; // uniform: every pair of this step is already converged
          PSD_COUNT(n_rot_steps);
          // A <- J' A J in ONE pass over 2x2 blocks (rows of pair P, columns of pair Q); V <- V J over (row, pair) items
          psd_update_pass<NB, 3>(A, A, V, rot_pq, rot_cs, items, ld);
          __syncthreads();
        }
        offmax = block_max(offmax, red);
        if (offmax <= thr) break;
      }
    };
    using std::integral_constant;
    if (pipelined && PIPE == 2) {
      switch (psd_blocks_per_lane(npairs, PSD_PIPE_THREADS, 3)) {
      case 1: sweeps_signal(integral_constant<int, 1>()); break;
      default: sweeps_signal(integral_constant<int, 2>()); break;
      }
    } else if (pipelined) {
      switch (psd_blocks_per_lane(npairs, PSD_PIPE_THREADS, PSD_PIPE_VR)) { // blocks of the upper triangle per update lane (K2 <= 72: at most 2)
      case 1: sweeps_pipelined(integral_constant<int, 1>()); break;
      default: sweeps_pipelined(integral_constant<int, 2>()); break;
      }
    } else if (PIPE) { // only K2 = 2 comes here in the pipelined instantiation
      sweeps_two_phase(integral_constant<int, 1>());
    } else {
      switch (psd_blocks_per_lane(npairs, PSD_THREADS)) { // uniform over the workgroup (K2 <= 92: 46 * 47 / 2 = 1081 <= 3 * 512)
      case 1: sweeps_two_phase(integral_constant<int, 1>()); break;
      case 2: sweeps_two_phase(integral_constant<int, 2>()); break;
      default: sweeps_two_phase(integral_constant<int, 3>()); break;
      }
    }
  }
  PSD_CLK(clk3);
  if (sweep >= PSD_MAX_SWEEPS && tid == 0) atomicAdd(status, 1); // did not converge: counted, not fatal (cones.c:1031-1032)
  // W = V diag(sqrt(max(lambda, 0)))   (cones.c:1036-1044); lambda = diag(A)
  A = Acur;
  __syncthreads();
  for (int e = tid; e < K2 * K2; e += PSD_THREADS) {
    const int i = use_lds ? e / K2 : e % K2, cidx = use_lds ? e % K2 : e / K2;
    const real lam = A[MI(cidx, cidx)];
    const real vv = V[MI(i, cidx)];
    if (Vg) Vg[i * ld + cidx] = vv; // the eigenbasis, for the next projection of this cone
    V[MI(i, cidx)] = vv * ((cidx < k && lam > (real)0) ? sqrt(lam) : (real)0);
  }
  __syncthreads();
  // X+ = W W', lower triangle only, repack with diagonal / sqrt(2)  (cones.c:1052-1063)
  const real inv_sqrt2 = (real)1 / sqrt2;
  if (cplx) { // Hermitian repack (cones.c:1139-1145): Re = (W W')[r][c], Im = (W W')[r+nn][c]
    for (int e = tid; e < nn * nn; e += PSD_THREADS) {
      int c = 0, rem = e;
      while (rem >= 2 * (nn - c) - 1) {
        rem -= 2 * (nn - c) - 1;
        ++c;
      }
      int ra, rb = c;
      real scale = 1;
      if (rem == 0) {
        ra = c;
        scale = inv_sqrt2;
      } else {
        const int r = c + 1 + (rem - 1) / 2;
        ra = ((rem - 1) & 1) ? r + nn : r;
      }
      real acc = 0;
      for (int cidx = 0; cidx < k; ++cidx) acc += V[MI(ra, cidx)] * V[MI(rb, cidx)];
      X[e] = acc * scale;
    }
    return;
  }
#ifndef SFLOAT
  // fp64 matrix cores: D(16x16) += A(16x4) B(4x16), v_mfma_f64_16x16x4_f64.  Lane l holds
  // A[l&15][l>>4], B[l>>4][l&15]; D element (row = (l>>4) + 4*reg, col = l&15), reg 0..3.
  {
    const int wave = tid >> 6, lane = tid & 63, nw = PSD_THREADS >> 6;
    const int T = (k + 15) >> 4; // 16x16 tiles per dimension
    const int ksteps = (K2 + 3) >> 2;
    const int li = lane & 15, lk = lane >> 4;
    for (int t = wave; t < T * T; t += nw) {
      const int ti = t / T, tj = t % T;
      if (tj > ti) continue; // lower triangle of tiles
      f64x4 acc = {0, 0, 0, 0};
      const int ra = ti * 16 + li, rb = tj * 16 + li;
      for (int ks = 0; ks < ksteps; ++ks) {
        const int kc = ks * 4 + lk;
        const double av = (ra < k && kc < K2) ? V[MI(ra, kc)] : 0.0;
        const double bv = (rb < k && kc < K2) ? V[MI(rb, kc)] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * 16 + lk + 4 * r, j = tj * 16 + li;
        if (i < k && j <= i) X[packed_index(i, j, k)] = i == j ? acc[r] * inv_sqrt2 : acc[r];
      }
    }
  }
#else
  {
    const int ntri = k * (k + 1) / 2;
    for (int e = tid; e < ntri; e += PSD_THREADS) {
      int j = 0, rem = e;
      while (rem >= k - j) {
        rem -= k - j;
        ++j;
      }
      const int i = j + rem;
      real acc = 0;
      for (int cidx = 0; cidx < k; ++cidx) acc += V[MI(i, cidx)] * V[MI(j, cidx)];
      if (i == j) acc *= inv_sqrt2;
      X[e] = acc;
    }
  }
#endif
#ifdef SCSAMD_PSD_CLOCKS
  __syncthreads();
  if (cone == 0 && (tid & 63) == 0 && (tid == 0 || tid == PSD_THREADS - 64))
    printf("PSDWAVE %s work %lld wait %lld\n", tid == 0 ? "update" : "lookahead", clk_work, clk_wait);
  if (cone == 0 && tid == 0)
    printf("PSDCLK pipe %d k %d unpack_warm %lld fro %lld sweeps %lld tail %lld nsweep %d steps %d rot_steps %d\n", PIPE, k, clk1 - clk0,
           clk2 - clk1, clk3 - clk2, (long long)clock64() - clk3, sweep, n_steps, n_rot_steps);
#endif
}

} // namespace scsamd
#include "psd_big.h"
namespace scsamd {

ConeDev::~ConeDev() { delete psd_big; }
void ConeDev::reset_warm_start() {
  psd_calls = 0;
  if (psd_big) psd_big->reset_warm_start();
}

// ----------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------
long long cone_total_rows(const ScsCone *k) {
  long long t = (long long)k->z + k->l + k->bsize;
  for (int i = 0; i < k->qsize; ++i) t += k->q[i];
  for (int i = 0; i < k->ssize; ++i) t += (long long)k->s[i] * (k->s[i] + 1) / 2;
  for (int i = 0; i < k->cssize; ++i) t += (long long)k->cs[i] * k->cs[i];
  t += 3LL * (k->ep + k->ed) + 3LL * k->psize;
  return t;
}

// mirrors the checks of reference src/cones.c:583-700 (validate_cones); every cone type of the default
// reference build is carried (zero, nonneg, box, SOC, PSD, complex PSD, exp, dual exp, power)
int validate_cone(const ScsCone *k, int m, bool verbose) {
#define CONE_FAIL(msg)                                                                             \
  do {                                                                                             \
    if (verbose) printf("%s\n", msg);                                                              \
    return -1;                                                                                     \
  } while (0)
  if (!k) CONE_FAIL("cone struct is NULL");
  if (k->z < 0) CONE_FAIL("free cone dimension error");
  if (k->l < 0) CONE_FAIL("lp cone dimension error");
  if (k->bsize < 0) CONE_FAIL("box cone dimension error");
  if (k->bsize > 1 && (!k->bl || !k->bu)) CONE_FAIL("box cone bounds missing");
  for (int i = 0; i < k->bsize - 1; ++i) {
    // NaN bounds, and infinities pointing the wrong way (bl = +inf / bu = -inf), are errors;
    // bl = -inf and bu = +inf are the valid one-sided cases
    if (k->bl[i] != k->bl[i] || k->bu[i] != k->bu[i]) CONE_FAIL("box cone error, bounds must not be NaN");
    if (k->bl[i] == (real)INFINITY || k->bu[i] == (real)-INFINITY)
      CONE_FAIL("box cone error, infinite bound in the wrong direction");
    if (k->bl[i] > k->bu[i]) CONE_FAIL("infeasible: box lower bound larger than upper bound");
  }
  if (k->qsize < 0 || (k->qsize > 0 && !k->q)) CONE_FAIL("soc cone dimension error");
  for (int i = 0; i < k->qsize; ++i)
    if (k->q[i] < 0) CONE_FAIL("soc cone dimension error");
  if (k->ssize < 0 || (k->ssize > 0 && !k->s)) CONE_FAIL("sd cone dimension error");
  for (int i = 0; i < k->ssize; ++i) {
    if (k->s[i] < 0) CONE_FAIL("sd cone dimension error");
  }
  if (k->cssize < 0 || (k->cssize > 0 && !k->cs)) CONE_FAIL("complex psd cone dimension error");
  for (int i = 0; i < k->cssize; ++i) {
    if (k->cs[i] < 0) CONE_FAIL("complex psd cone dimension error");
    if (2 * k->cs[i] > PSD_K_LIMIT) CONE_FAIL("complex sd cone larger than 512 x 512 not supported by the MI355X backend");
  }
  if (k->ep < 0) CONE_FAIL("ep cone dimension error");
  if (k->ed < 0) CONE_FAIL("ed cone dimension error");
  if (k->psize < 0 || (k->psize > 0 && !k->p)) CONE_FAIL("power cone dimension error");
  for (int i = 0; i < k->psize; ++i)
    if (!std::isfinite((double)k->p[i]) || k->p[i] < -1 || k->p[i] > 1)
      CONE_FAIL("power cone error, values must be finite and in [-1,1]");
  if (cone_total_rows(k) != m) {
    if (verbose)
      printf("cone dimensions %lld not equal to num rows in A = m = %d\n", cone_total_rows(k), m);
    return -1;
  }
  return 0;
#undef CONE_FAIL
}

static inline int small_grid(long long len) {
  long long g = (len + SCSAMD_BLOCK - 1) / SCSAMD_BLOCK;
  return (int)std::max<long long>(1, std::min<long long>(g, 2048));
}

void ConeDev::init(const ScsCone *k, int m_, const real *D, hipStream_t s) {
  m = m_;
  stream = s;
  z = k->z;
  l = k->l;
  bsize = k->bsize;
  box_off = z + l;
  int off = z + l;
  if (bsize > 1) {
    // normalize_box_cone (cones.c:1161-1177) applied to a private copy.  The reference only calls it when a
    // scaling exists (cones.c:1561: `if (scal)`), so without one the bounds are used exactly as given -- a
    // "1e20 = infinite" bound then stays the finite number 1e20, as in the reference
    std::vector<real> hl(bsize - 1), hu(bsize - 1);
    const real *Db = D ? D + box_off : nullptr;
    for (int j = 0; j < bsize - 1; ++j) {
      if (!Db) {
        hu[j] = k->bu[j];
        hl[j] = k->bl[j];
        continue;
      }
      const real f = Db[j + 1] / Db[0];
      hu[j] = k->bu[j] >= (real)MAX_BOX_VAL ? (real)INFINITY : k->bu[j] * f;
      hl[j] = k->bl[j] <= (real)-MAX_BOX_VAL ? (real)-INFINITY : k->bl[j] * f;
    }
    bl.alloc(bsize - 1);
    bu.alloc(bsize - 1);
    bl.upload(hl.data(), bsize - 1, stream);
    bu.upload(hu.data(), bsize - 1, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  box_t.alloc(1);
  box_multi = bsize - 1 >= BOX_MULTI_MIN;
  if (const char *e = opt_get("box_multi")) box_multi = atoi(e) != 0 && bsize > 1; // tests force either path
  if (box_multi) {
    box_part.alloc((size_t)4 * BOX_MULTI_GRID);
    box_ctl.alloc(8); // one BoxCtl
  }
  {
    const real one = 1; // cones.c:1560
    box_t.upload(&one, 1, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  off += bsize;
  // second-order cones
  std::vector<int> toff, tlen, boff, blen, bt0, tcone, tileoff, tilelen;
  for (int i = 0; i < k->qsize; ++i) {
    const int q = k->q[i];
    if (q <= SOC_TINY_MAX) {
      toff.push_back(off);
      tlen.push_back(q);
    } else {
      const int c = (int)boff.size();
      boff.push_back(off);
      blen.push_back(q);
      bt0.push_back((int)tcone.size());
      for (int o = 0; o < q; o += SOC_TILE) {
        tcone.push_back(c);
        tileoff.push_back(off + o);
        tilelen.push_back(std::min(SOC_TILE, q - o));
      }
    }
    off += q;
  }
  bt0.push_back((int)tcone.size());
  n_tiny = (int)toff.size();
  n_big = (int)boff.size();
  n_tiles = (int)tcone.size();
  auto up = [&](DevBuf<int> &d, std::vector<int> &h) {
    d.alloc(h.size() ? h.size() : 1);
    if (h.size()) d.upload(h.data(), h.size(), stream);
  };
  up(tiny_off, toff);
  up(tiny_len, tlen);
  up(big_off, boff);
  up(big_len, blen);
  up(big_tile0, bt0);
  up(tile_cone, tcone);
  up(tile_off, tileoff);
  up(tile_len, tilelen);
  tile_part.alloc(n_tiles ? n_tiles : 1);
  big_coef.alloc(n_big ? 2 * (size_t)n_big : 2);
  // PSD
  std::vector<int> poff, pk;
  psd_kmax = 0;
  for (int i = 0; i < k->ssize; ++i) {
    poff.push_back(off);
    pk.push_back(k->s[i]);
    psd_kmax = std::max(psd_kmax, (int)k->s[i]);
    off += k->s[i] * (k->s[i] + 1) / 2;
  }
  for (int i = 0; i < k->cssize; ++i) { // complex PSD cones follow the real ones (cones.c:1396-1404)
    poff.push_back(off);
    pk.push_back(-2 * k->cs[i]);
    psd_kmax = std::max(psd_kmax, 2 * (int)k->cs[i]);
    off += k->cs[i] * k->cs[i];
  }
  n_psd = (int)poff.size();
  up(psd_off, poff);
  up(psd_k, pk);
  psd_lds_kmax = 0;
  for (int kk : pk) {
    const int ka = kk < 0 ? -kk : kk;
    if (ka <= PSD_LDS_KMAX) psd_lds_kmax = std::max(psd_lds_kmax, ka);
  }
  delete psd_big;
  psd_big = nullptr;
  if (n_psd && psd_kmax > PSD_LDS_KMAX) { // blocks that do not fit one CU's LDS: chip-wide Jacobi steps (psd_big.h)
    psd_big = new BigPsd;
    psd_big->init(pk, PSD_LDS_KMAX, stream);
  }
  psd_calls = 0;
  psd_pipe = 1;
  if (const char *e = opt_get("psd_pipe")) psd_pipe = std::max(0, std::min(2, atoi(e)));
  // warm start of the LDS kernel: sized and gated by the largest block that kernel handles (blocks beyond the LDS path
  // carry their own basis in psd_big, and must not switch the small blocks' warm start off)
  psd_vprev.release();
  psd_tscratch.release();
  int warm_kmax = PSD_LDS_KMAX; // round 4: every order of the LDS kernel is warm started (SCS_AMD_PSD_WARM_KMAX=72 restores round 3's gate)
  if (const char *e = opt_get("psd_warm_kmax")) warm_kmax = std::max(0, std::min(PSD_LDS_KMAX, atoi(e)));
  if (n_psd && psd_lds_kmax >= 2 && psd_lds_kmax <= warm_kmax && !opt_get("psd_cold")) {
    const size_t per_cone = (size_t)((psd_lds_kmax + 1) & ~1) * (((psd_lds_kmax + 1) & ~1) | 1);
    psd_vprev.alloc((size_t)n_psd * per_cone);
    if (psd_lds_kmax > PSD_WARM_KMAX) psd_tscratch.alloc((size_t)n_psd * per_cone); // T = A Vp of the warm start does not fit LDS
  }
  ep = k->ep;
  ed = k->ed;
  psize = k->psize;
  exp_off = off;
  off += 3 * (ep + ed + psize);
  pow_a.alloc(psize > 0 ? psize : 1);
  if (psize > 0) pow_a.upload(k->p, psize, stream);
  status.alloc(1);
  hstatus.alloc(1);
  HIP_CHECK(hipStreamSynchronize(stream));
  if (off != m) throw HipError("scs_amd: cone rows do not add up to m");
}

void ConeDev::proj_primal(real *cw, const real *r_y) {
  if (z + l > 0)
    hipLaunchKernelGGL(k_zero_pos, dim3(small_grid(z + l)), dim3(SCSAMD_BLOCK), 0, stream, cw, z, l);
  if (bsize > 0) {
    const real *rb = r_y ? r_y + box_off : (const real *)nullptr;
    if (box_multi) {
      const int g = std::max(1, std::min(BOX_MULTI_GRID, (bsize - 1 + 8 * SCSAMD_BLOCK - 1) / (8 * SCSAMD_BLOCK)));
      BoxCtl *ctl = reinterpret_cast<BoxCtl *>(box_ctl.p);
      for (int it = 0; it <= BOX_MAX_ITERS; ++it)
        hipLaunchKernelGGL(k_box_step, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, cw + box_off, bl.p, bu.p, bsize, box_t.p,
                           rb, ctl, box_part.p, it);
      hipLaunchKernelGGL(k_box_apply, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, cw + box_off, bl.p, bu.p, bsize, box_t.p,
                         ctl);
    } else {
      hipLaunchKernelGGL(k_box, dim3(1), dim3(BOX_THREADS), 0, stream, cw + box_off, bl.p, bu.p, bsize, box_t.p, rb);
    }
  }
  if (n_tiny)
    hipLaunchKernelGGL(k_soc_tiny, dim3(small_grid(n_tiny)), dim3(SCSAMD_BLOCK), 0, stream, cw, tiny_off.p,
                       tiny_len.p, n_tiny);
  if (n_big) {
    const int g = std::min(n_tiles, 8192);
    hipLaunchKernelGGL(k_soc_tile_partial, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, cw, tile_off.p,
                       tile_len.p, tile_cone.p, big_off.p, tile_part.p, n_tiles);
    hipLaunchKernelGGL(k_soc_finalize, dim3(small_grid(n_big)), dim3(SCSAMD_BLOCK), 0, stream, cw,
                       big_off.p, big_tile0.p, tile_part.p, big_coef.p, n_big);
    hipLaunchKernelGGL(k_soc_tile_apply, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, cw, tile_off.p, tile_len.p,
                       tile_cone.p, big_off.p, big_coef.p, n_tiles);
  }
  if (n_psd) {
    const int lds_kmax = std::min(psd_kmax, psd_lds_kmax); // largest block order held in LDS
    const int K2l = (lds_kmax + 1) & ~1;
    const bool carry = psd_vprev.p != nullptr;
    // round 5: pipelined step (second copy of A in LDS) whenever three matrices fit; SCS_AMD_PSD_PIPE=0 keeps the two-phase step (A/B)
    const int pipe = K2l <= PSD_WARM_KMAX ? psd_pipe : 0; // psd_pipe: option read in init (0 two-phase, 1 look-ahead, 2 signal form)
    const size_t lds = PSD_LDS_HEADER + (size_t)((pipe || (carry && !psd_tscratch.p)) ? 3 : 2) * K2l * (K2l | 1) * sizeof(real);
    const int warm = carry && (psd_calls % PSD_WARM_RESET) != 0;
    ++psd_calls;
    auto launch = [&](auto kern) {
      if (lds > 48 * 1024)
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3(n_psd), dim3(PSD_THREADS), lds, stream, cw, psd_off.p, psd_k.p, psd_tscratch.p, lds_kmax, lds_kmax,
                         status.p, psd_vprev.p, warm);
    };
    if (pipe == 2) launch(k_psd_jacobi<2>);
    else if (pipe == 1) launch(k_psd_jacobi<1>);
    else launch(k_psd_jacobi<0>);
    if (psd_big) psd_big->project(cw, psd_off.p, psd_k.p, status.p, stream);
  }
  proj_exp_pow(cw);
}

// number of PSD block projections that hit the sweep cap since the last call; resets the device counter
int ConeDev::take_status(hipStream_t st) {
  HIP_CHECK(hipMemcpyAsync(hstatus.p, status.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  const int h = hstatus.p[0];
  if (h) HIP_CHECK(hipMemsetAsync(status.p, 0, sizeof(int), st));
  return h;
}

void ConeDev::proj_exp_pow(real *cw) {
  if (ep + ed + psize > 0)
    hipLaunchKernelGGL(k_exp_pow, dim3(small_grid(ep + ed + psize)), dim3(SCSAMD_BLOCK), 0, stream, cw + exp_off, ep, ed,
                       psize, pow_a.p);
}

void ConeDev::proj_dual(real *x, real *scratch, const real *r_y) {
  const int g = small_grid(m);
  hipLaunchKernelGGL(k_moreau_pre, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, x, scratch, r_y, m);
  proj_primal(x, r_y);
  hipLaunchKernelGGL(k_moreau_post, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, x, scratch, r_y, m);
}

} // namespace scsamd

// ============================================================================
// B1': host-pointer cone projection (replaces _scs_init_cone / _scs_proj_dual_cone /
// _scs_finish_cone, reference include/cones.h:80-90)
// ============================================================================
using namespace scsamd;

struct SCS_AMD_CONE_WORK {
  ConeDev cd;
  hipStream_t stream = nullptr;
  ~SCS_AMD_CONE_WORK() {
    if (stream) (void)hipStreamDestroy(stream);
  }
};

extern "C" {

ScsAmdConeWork *scs_amd_cone_init(const ScsCone *k, scs_int m, const scs_float *D) {
  if (validate_cone(k, m, true) < 0) return nullptr;
  ScsAmdConeWork *c = nullptr;
  try {
    c = new ScsAmdConeWork();
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->cd.init(k, m, D, c->stream);
    c->cd.x_stage.alloc(m);
    c->cd.s_stage.alloc(m);
    c->cd.r_stage.alloc(m);
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    delete c;
    return nullptr;
  }
  return c;
}

scs_int scs_amd_cone_proj_dual(ScsAmdConeWork *c, scs_float *x, const scs_float *r_y) {
  if (!c || !x) return -1;
  try {
    ConeDev &cd = c->cd;
    cd.x_stage.upload(x, cd.m, c->stream);
    if (r_y) cd.r_stage.upload(r_y, cd.m, c->stream);
    cd.proj_dual(cd.x_stage.p, cd.s_stage.p, r_y ? cd.r_stage.p : nullptr);
    cd.x_stage.download(x, cd.m, c->stream);
    const int bad = cd.take_status(c->stream); // synchronises the stream
    HIP_CHECK(hipGetLastError());
    if (bad > 0) return 1; // like LAPACK info > 0 in the reference: positive, not fatal (src/cones.c:1031-1032)
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return 0;
}

void scs_amd_cone_finish(ScsAmdConeWork *c) { delete c; }

} // extern "C"
