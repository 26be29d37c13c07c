// psd_big.h -- PSD blocks whose A and V do not fit one CU's LDS (order > PSD_LDS_KMAX): the same parallel cyclic
// Jacobi iteration as k_psd_jacobi (cones.hip), every step spread over the whole chip instead of one workgroup.
// Replaces LAPACK dsyevr / zheevr + the reconstruction of reference src/cones.c:999-1155 for those blocks; no size
// limit other than memory.  Included by cones.hip only (uses its RotCS, packed_index, psd_unpack_entry).
//
// Per big block (all in HBM, column-major, common leading dimension ld = even-padded largest order):
//   two copies of A (K2 x K2 working matrix), V (eigenvectors), the carried eigenbasis, a small control record.
// What ships (rounds 3-4) is the BLOCKED iteration further down: a tournament over 32-wide block columns, one fused launch per outer
// step (k_bj_fused: the 64 x 64 subproblems of step k + 1 rotated beside the matrix-core update of step k), the cross sweep software-
// pipelined (bj_inner_sweep_cross).  The single-column form described next is round 2's, kept behind SCS_AMD_PSD_BLOCKED=0 as the
// baseline of the measurements, and shares the control record, the sweep bookkeeping, the warm start and the reconstruction with it.
// One Jacobi step = ONE launch over all big blocks (blockIdx.y = block), k_bp_step: A is double-buffered (the step reads
// the matrix as it stood when the step began and writes the other copy), so every workgroup can form the rotations it
// needs straight from A without racing the workgroups that are rewriting those entries -- no separate "parameters"
// launch and no tables.  A workgroup owns a 16 x 16 tile of (row pair, column pair) 2x2 blocks (lanes 0-31 form its 32
// rotations once, everybody reads them from LDS) or a 64-row x 4-pair tile of V; same formulas, same threshold rule,
// same round-robin order, same exact zero for a rotated pair's own entry as the LDS kernel.  The running maximum of the
// off-diagonal entries met in a sweep is an atomic max (order-independent, so still deterministic).
// k_bp_sweep_end closes a sweep (convergence, sweep cap as in cones.c:1031), the host reads one int per sweep.
// A step costs ~5 us however small the block (a kernel that hands data to the next one), so the time is steps x sweeps;
// the earlier two-launch form (parameters, then update) took ~10 us per step.
// Reconstruction X+ = W W' (W = V diag(sqrt(max(lambda, 0)))) runs on the fp64 matrix cores over all CUs (k_bp_gram).
// Everything is deterministic: no atomics in sums (the one atomic is a max), one workgroup owns every reduction.
//
// Measured (profiles/r2_bench_psd_sizes_*.jsonl, r3_*, r4_psd_fused_step.md): see DESIGN.md section 6.
#pragma once

namespace scsamd {

struct BigPsdCtl {
  real thr, fro;
  unsigned long long offmax_bits[2]; // bit pattern of the largest |a_pq| met in a sweep (non-negative: orders like the value), by sweep parity:
                                     // the fused step (k_bj_fused) runs the first inner sweep of sweep s + 1 before sweep s is closed
  unsigned long long left_bits;   // the same of the largest off-diagonal entry of the matrix as the sweep just closed LEFT it (k_bp_offscan)
  int cur[2];                     // which A copy holds the block before launch g: cur[g & 1] (written for g+1 by the launch itself)
  int done, sweeps, kraw; // kraw: signed order (negative = complex embedding), copied here so that a step kernel
                          // needs ONE dependent read (this record) before it touches A
};

constexpr int BP_THREADS = 256;
constexpr int BP_PARAM_THREADS = 1024;

struct BigPsdView {
  int nbig, ld;           // blocks, common leading dimension (= K2 of the largest)
  const int *id;          // index of each big block in psd_off / psd_k
  const int *psd_off, *psd_k;
  real *A, *V;            // nbig * ld * ld each (A: copy 0 of the working matrix)
  real *A1;               // copy 1 of the working matrix
  BigPsdCtl *ctl;         // nbig
};

constexpr int BJ_B = 32;       // blocked iteration: width of a block column
constexpr int BJ_W = 2 * BJ_B; // order of the subproblem of a pair of block columns

struct BlockShape {
  int k, K2, nn, npairs;
  int K64, nbc; // blocked iteration: order padded to whole pairs of block columns, number of block columns (even)
  bool cplx;
};
__device__ __forceinline__ BlockShape bp_shape_raw(int kraw) {
  BlockShape s;
  s.cplx = kraw < 0;
  s.k = s.cplx ? -kraw : kraw;
  s.nn = s.k / 2;
  s.K2 = (s.k + 1) & ~1;
  s.npairs = s.K2 / 2;
  s.K64 = (s.k + BJ_W - 1) / BJ_W * BJ_W;
  s.nbc = s.K64 / BJ_B;
  return s;
}
__device__ __forceinline__ BlockShape bp_shape(const BigPsdView &B, int b) { return bp_shape_raw(B.psd_k[B.id[b]]); }

// A <- unpacked block (diagonal * sqrt 2), V <- I
__global__ __launch_bounds__(BP_THREADS) void k_bp_unpack(BigPsdView B, const real *__restrict__ x, int set_identity, int blocked) {
  const int b = blockIdx.y;
  const BlockShape s = bp_shape(B, b);
  const real *X = x + B.psd_off[B.id[b]];
  real *A = B.A + (size_t)b * B.ld * B.ld, *V = B.V + (size_t)b * B.ld * B.ld;
  const int span = blocked ? s.K64 : s.K2; // the blocked iteration works on whole pairs of block columns: zero rows / columns, unit diagonal in V
  const long long total = (long long)span * span;
  for (long long e = (long long)blockIdx.x * BP_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * BP_THREADS) {
    const int i = (int)(e % span), j = (int)(e / span);
    A[(size_t)j * B.ld + i] = psd_unpack_entry(X, s.k, s.cplx, s.nn, i, j);
    if (set_identity) V[(size_t)j * B.ld + i] = i == j ? (real)1 : (real)0;
  }
}

// Frobenius norm, threshold, control record armed.  Two launches with a fixed summation order (round 6): BP_NORM_G workgroups per block
// sum whole columns (column j goes to workgroup j mod BP_NORM_G; coalesced, no index division) into one partial each, one workgroup
// per block adds the partials in their order.  (Rounds 2-5: ONE workgroup per block walked the whole matrix with a 64-bit division per
// entry -- 31 us at order 256, 407 us at order 1024, 4 % of that projection.)
constexpr int BP_NORM_G = 64;
__global__ __launch_bounds__(BP_THREADS) void k_bp_norm_part(BigPsdView B, real *__restrict__ part) {
  __shared__ real red[BP_THREADS / SCSAMD_WAVE];
  const int b = blockIdx.y, g = blockIdx.x;
  const BlockShape s = bp_shape(B, b);
  const real *A = B.A + (size_t)b * B.ld * B.ld;
  real fro = 0;
  for (int j = g; j < s.K2; j += BP_NORM_G) {
    const real *col = A + (size_t)j * B.ld;
    for (int i = threadIdx.x; i < s.K2; i += BP_THREADS) {
      const real v = col[i];
      fro += v * v;
    }
  }
  fro = block_sum(fro, red);
  if (threadIdx.x == 0) part[(size_t)b * BP_NORM_G + g] = fro;
}
__global__ __launch_bounds__(SCSAMD_WAVE) void k_bp_norm(BigPsdView B, const real *__restrict__ part) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  const BlockShape s = bp_shape(B, b);
  real fro = 0;
  for (int g = 0; g < BP_NORM_G; ++g) fro += part[(size_t)b * BP_NORM_G + g];
  fro = sqrt(fro);
  const real eps = sizeof(real) == 8 ? (real)1e-15 : (real)1e-7;
  BigPsdCtl *c = B.ctl + b;
  c->fro = fro;
  // same rule as k_psd_jacobi (fp32: not below the rounding noise of the rotations)
  c->thr = sizeof(real) == 8 ? eps * fro / (real)s.k : fmaxf(eps * fro / (real)s.k, (real)2.4e-7 * fro);
  c->offmax_bits[0] = c->offmax_bits[1] = 0ull;
  c->left_bits = 0ull;
  c->cur[0] = c->cur[1] = 0;
  c->done = fro > (real)0 ? 0 : 1;
  c->sweeps = 0;
  c->kraw = B.psd_k[B.id[b]];
}

__device__ __forceinline__ unsigned long long bp_bits(real v) { // v >= 0
  return sizeof(real) == 8 ? (unsigned long long)__double_as_longlong((double)v) : (unsigned long long)__float_as_uint((float)v);
}
__device__ __forceinline__ real bp_from_bits(unsigned long long b) {
  return sizeof(real) == 8 ? (real)__longlong_as_double((long long)b) : (real)__uint_as_float((unsigned)b);
}

// rotation of pair i of round-robin step `step`, from the matrix as it stands before the step
__device__ __forceinline__ void bp_rotation(const real *Aold, size_t ld, int i, int step, int K2, int k, real thr, int2 &pq,
                                            RotCS &cs, real &aa_out) {
  int p = i == 0 ? 0 : 1 + ((i - 1 + step) % (K2 - 1));
  int q = 1 + ((K2 - 2 - i + step) % (K2 - 1));
  if (p > q) {
    const int t = p;
    p = q;
    q = t;
  }
  real c = 1, s = 0;
  const real apq = Aold[q * ld + p], aqq = Aold[q * ld + q], app = Aold[p * ld + p];
  const real aa = absval(apq);
  aa_out = q < k ? aa : (real)0;
  if (q < k && aa > thr) {
    // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), written without the first division
    const real d = aqq - app, bb = (real)2 * apq;
    jacobi_cs(d, bb, c, s);
  }
  pq = make_int2(p, q);
  cs = RotCS{c, s};
}

constexpr int BP_TILE = 16;   // a workgroup's tile of 2x2 blocks: 16 row pairs x 16 column pairs
constexpr int BP_VROWS = 64;  // V tile: 64 rows x 4 pairs
constexpr int BP_VPAIRS = BP_THREADS / BP_VROWS;

// arg = (launch parity) | (round-robin step << 1): the control record's `cur` is double-buffered by LAUNCH parity (a sweep
// of the largest block has an odd or even number of steps), the pairing follows the step
__global__ __launch_bounds__(BP_THREADS) void k_bp_step(BigPsdView B, int arg) {
  const int slot = arg & 1, step = arg >> 1;
  __shared__ int2 s_pq[2 * BP_TILE];
  __shared__ RotCS s_cs[2 * BP_TILE];
  const int b = blockIdx.y, tid = threadIdx.x;
  BigPsdCtl *ctl = B.ctl + b;
  const int done = ctl->done, kraw = ctl->kraw, cur = ctl->cur[slot];
  const real thr = ctl->thr;
  const BlockShape sh = bp_shape_raw(kraw);
  const bool active = !done && step < sh.K2 - 1; // a block smaller than the largest has fewer steps per sweep
  if (blockIdx.x == 0 && tid == 0) ctl->cur[slot ^ 1] = active ? cur ^ 1 : cur;
  if (!active) return;
  const size_t ld = B.ld, mat = (size_t)b * ld * ld;
  const real *Aold = (cur ? B.A1 : B.A) + mat;
  real *Anew = (cur ? B.A : B.A1) + mat;
  real *V = B.V + mat;
  const int npairs = sh.npairs, K2 = sh.K2, k = sh.k;
  const int TP = (npairs + BP_TILE - 1) / BP_TILE;
  const int nta = TP * TP;
  const int TQv = (npairs + BP_VPAIRS - 1) / BP_VPAIRS, TR = (K2 + BP_VROWS - 1) / BP_VROWS;
  int tile = blockIdx.x;
  if (tile < nta) {
    // ---- A <- J' A J on a 16 x 16 tile of 2x2 blocks: rows of pairs [16 tp, +16), columns of pairs [16 tq, +16)
    const int tq = tile / TP, tp = tile % TP;
    if (tid < 2 * BP_TILE) {
      const int i = tid < BP_TILE ? BP_TILE * tp + tid : BP_TILE * tq + (tid - BP_TILE);
      int2 pq = make_int2(0, 0);
      RotCS cs{(real)1, (real)0};
      real aa = 0;
      if (i < npairs) bp_rotation(Aold, ld, i, step, K2, k, thr, pq, cs, aa);
      s_pq[tid] = pq;
      s_cs[tid] = cs;
      // every pair is a "row pair" of exactly the tiles with tq == 0: they carry the sweep's off-diagonal maximum
      if (tq == 0 && tid < BP_TILE) {
        real mx = aa;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          const real w = __shfl_down(mx, o, 64);
          mx = w > mx ? w : mx;
        }
        if (tid == 0 && mx > (real)0) atomicMax(&ctl->offmax_bits[ctl->sweeps & 1], bp_bits(mx));
      }
    }
    __syncthreads();
    const int lp = tid & (BP_TILE - 1), lq = tid >> 4;
    const int P = BP_TILE * tp + lp, Q = BP_TILE * tq + lq;
    if (P < npairs && Q < npairs) {
      // consecutive lanes walk the ROW pairs: consecutive p1 (and q1) -> consecutive addresses in column-major storage
      const int2 pq1 = s_pq[lp], pq2 = s_pq[BP_TILE + lq];
      const RotCS r1 = s_cs[lp], r2 = s_cs[BP_TILE + lq];
      const int p1 = pq1.x, q1 = pq1.y, p2 = pq2.x, q2 = pq2.y;
      const real c1 = r1.c, s1 = r1.s, c2 = r2.c, s2 = r2.s;
      const size_t i11 = p2 * ld + p1, i12 = q2 * ld + p1, i21 = p2 * ld + q1, i22 = q2 * ld + q1;
      const real a11 = Aold[i11], a12 = Aold[i12], a21 = Aold[i21], a22 = Aold[i22];
      const real r11 = c1 * a11 - s1 * a21, r12 = c1 * a12 - s1 * a22;
      const real r21 = s1 * a11 + c1 * a21, r22 = s1 * a12 + c1 * a22;
      const bool own = P == Q && s1 != (real)0; // the rotated pair's own off-diagonal entry: exact zero
      Anew[i11] = c2 * r11 - s2 * r12;
      Anew[i12] = own ? (real)0 : s2 * r11 + c2 * r12;
      Anew[i21] = own ? (real)0 : c2 * r21 - s2 * r22;
      Anew[i22] = s2 * r21 + c2 * r22;
    }
    return;
  }
  tile -= nta;
  if (tile >= TQv * TR) return;
  // ---- V <- V J on a tile of 64 rows x 4 pairs (in place: a lane owns its two entries)
  const int tqv = tile / TR, tr = tile % TR;
  if (tid < BP_VPAIRS) {
    const int i = BP_VPAIRS * tqv + tid;
    int2 pq = make_int2(0, 0);
    RotCS cs{(real)1, (real)0};
    real aa = 0;
    if (i < npairs) bp_rotation(Aold, ld, i, step, K2, k, thr, pq, cs, aa);
    s_pq[tid] = pq;
    s_cs[tid] = cs;
  }
  __syncthreads();
  const int li = tid & (BP_VROWS - 1), lq = tid >> 6;
  const int row = BP_VROWS * tr + li, Q = BP_VPAIRS * tqv + lq;
  if (row < K2 && Q < npairs) {
    const int2 pq2 = s_pq[lq];
    const RotCS r2 = s_cs[lq];
    if (r2.s != (real)0) {
      const size_t ip = pq2.x * ld + row, iq = pq2.y * ld + row;
      const real vp = V[ip], vq = V[iq];
      V[ip] = r2.c * vp - r2.s * vq;
      V[iq] = r2.s * vp + r2.c * vq;
    }
  }
}

// =====================================================================================================================
// Blocked iteration: the round-robin tournament runs over BLOCK COLUMNS of width 32 instead of single
// columns.  One outer step pairs the block columns (I, J); for every pair
//   k_bj_inner   one workgroup pulls the 64 x 64 subproblem S = A[IJ, IJ] into LDS, runs ONE sweep of the same parallel
//                cyclic Jacobi rotations as k_psd_jacobi on it (63 inner steps, same formulas, same threshold, exact zero
//                for a rotated pair's own entry) and leaves the accumulated orthogonal factor Q (64 x 64), the rotated
//                subproblem S' = Q' S Q and a "rotated at all" flag in HBM;
//   k_bj_update  applies the step to the whole matrix on the fp64 matrix cores: tile (P, Q) of the double-buffered A
//                becomes Q_P' A[P, Q] Q_Q (two 64^3 products per workgroup, operands staged in LDS with a leading
//                dimension of 66 so that the v_mfma_f64_16x16x4 operand reads are bank-conflict free), the diagonal tiles
//                take S' as the inner sweep left it (rotation arithmetic keeps its tiny entries relatively accurate and
//                its zeros exact; a GEMM would refill them with rounding noise of the diagonal's size), and the
//                eigenvector tiles V[:, Q] <- V[:, Q] Q_Q.
// A sweep is nbc - 1 outer steps of two launches each instead of K - 1 launches (1024 x 1024: 62 launches instead of 1023)
// and the O(K^3) work of a sweep moves onto the matrix cores.  Convergence (largest off-diagonal entry met during a sweep
// <= eps |A|_F / k), sweep cap, warm start and reconstruction are unchanged.  Padding rows / columns (order rounded up to a
// multiple of 64) are zero in A and unit in V; a rotation never touches them (q < k, as in the LDS kernel).
constexpr int BJ_LD = BJ_W + 2;        // LDS leading dimension of the product operands (66: rows 2 banks apart)
constexpr int BJ_ILD = BJ_W | 1;       // LDS leading dimension of the rotation sweep (65: row and column walks conflict free)
#ifndef BJ_INNER_THREADS_OVERRIDE
constexpr int BJ_INNER_THREADS = 512;
#else
constexpr int BJ_INNER_THREADS = BJ_INNER_THREADS_OVERRIDE;
#endif
constexpr int BJ_UPD_THREADS_MAX = 1024;                // largest workgroup that runs bj_inner_sweep (k_bj_fused)
constexpr int BJ_NBLK = BJ_B * BJ_B / BJ_INNER_THREADS; // 2x2 blocks of S per lane and inner step
constexpr int BJ_NROW = BJ_W * BJ_B / BJ_INNER_THREADS; // row pairs of Q per lane and inner step
static_assert(BJ_NBLK * BJ_INNER_THREADS == BJ_B * BJ_B && BJ_NROW * BJ_INNER_THREADS == BJ_W * BJ_B, "k_bj_inner: whole items per lane");
constexpr size_t BJ_INNER_LDS = (size_t)3 * BJ_W * BJ_ILD * sizeof(real) + 2 * BJ_B * sizeof(RotCS); // S (two copies), Q, two generations of (c, s)
constexpr int BJ_INNER_LAUNCH = 1024; // (= BJ_UPD_THREADS_MAX) the cross sweep's workgroup: fifteen worker waves and the wave that forms the rotations
constexpr size_t BJ_UPDATE_LDS = (size_t)4 * BJ_W * BJ_LD * sizeof(real);

__device__ __forceinline__ int2 bj_pair(int i, int step, int nbc) { // round robin over block columns, as over columns
  int I = i == 0 ? 0 : 1 + ((i - 1 + step) % (nbc - 1));
  int J = 1 + ((nbc - 2 - i + step) % (nbc - 1));
  if (I > J) {
    const int t = I;
    I = J;
    J = t;
  }
  return make_int2(I, J);
}
// The sweep's schedule (SCS_AMD_PSD_CROSS, default on): outer step 0 pairs the block columns (0,1), (2,3), ... and rotates only WITHIN
// each block column (two independent round robins over 32 indices, 31 inner steps); outer steps 1 .. nbc - 1 are the tournament over
// block columns and rotate only the CROSS pairs (p in I, q in J; 32 inner steps of the cyclic shift q = (i + st) mod 32): every index
// pair is rotated exactly once per sweep -- a cyclic Jacobi ordering -- instead of the within-block pairs once per outer step.
// The numpy emulation of this schedule needs 10 - 12 sweeps where the full 63-step subproblem sweep needs 8 - 10; an outer step costs
// half.  ostep = the `step` field of a launch: 0 = the within pass, s >= 1 = tournament step s - 1 (cross schedule); with the full
// schedule ostep = tournament step, no within pass.
__device__ __forceinline__ int2 bj_pair_sched(int i, int ostep, int nbc, int cross) {
  if (!cross) return bj_pair(i, ostep, nbc);
  if (ostep == 0) return make_int2(2 * i, 2 * i + 1);
  return bj_pair(i, ostep - 1, nbc);
}
__device__ __forceinline__ int bj_outer_steps(int nbc, int cross) { return cross ? nbc : nbc - 1; }
// pair i (0..31) of inner step st for the three kinds of inner sweep
__device__ __forceinline__ int2 bj_inner_pair(int i, int st, int kind) { // kind 0: full round robin over 64; 1: within; 2: cross
  int p, q;
  if (kind == 2) {
    p = i;
    q = BJ_B + ((i + st) & (BJ_B - 1));
  } else {
    const int n = kind == 1 ? BJ_B : BJ_W, j = kind == 1 ? (i & (BJ_B / 2 - 1)) : i, off = kind == 1 && i >= BJ_B / 2 ? BJ_B : 0;
    p = j == 0 ? 0 : 1 + ((j - 1 + st) % (n - 1));
    q = 1 + ((n - 2 - j + st) % (n - 1));
    if (p > q) {
      const int t = p;
      p = q;
      q = t;
    }
    p += off;
    q += off;
  }
  return make_int2(p, q);
}
template <int KIND> __device__ __forceinline__ int2 bj_inner_pair_k(int i, int st) { // the same with the kind known at compile time (no run-time modulus)
  if (KIND == 2) return make_int2(i, BJ_B + ((i + st) & (BJ_B - 1)));
  constexpr int n = KIND == 1 ? BJ_B : BJ_W;
  const int j = KIND == 1 ? (i & (BJ_B / 2 - 1)) : i, off = KIND == 1 && i >= BJ_B / 2 ? BJ_B : 0;
  int p = j == 0 ? 0 : 1 + ((j - 1 + st) % (n - 1));
  int q = 1 + ((n - 2 - j + st) % (n - 1));
  if (p > q) {
    const int t = p;
    p = q;
    q = t;
  }
  return make_int2(p + off, q + off);
}
// global row / column of local index l (0..63) of the pair (I, J)
__device__ __forceinline__ int bj_gidx(int2 IJ, int l) { return l < BJ_B ? IJ.x * BJ_B + l : IJ.y * BJ_B + (l - BJ_B); }

// The inner sweep proper, on the subproblem already in LDS (S[r * BJ_ILD + c]); Q <- I here.  Every thread of the workgroup calls it;
// threads beyond BJ_INNER_THREADS only keep the barriers company (the fused step runs it in a workgroup of BJ_UPD_THREADS).
// Leaves Q, S' and the "rotated at all" flag in HBM (Qg / Sg: 64 x 64 column-major) and the sweep's off-diagonal maximum in the
// control record (slot `offslot`).
// A step is two phases and two barriers: 32 lanes form the rotations (three LDS reads, fp64 rsqrt chain: ~840 clocks measured), then
// 1024 2x2 blocks of S and 2048 row pairs of Q over the lanes: every lane owns BJ_NBLK blocks and BJ_NROW row pairs and asks for
// everything it needs -- the (c, s) of its pairs, its operands, the "any rotation" flag -- in ONE batch of LDS reads: the pair indices
// are arithmetic (the kind of sweep is a template parameter), so no address depends on a table.  (With the pairs read from a table
// the phase was three dependent LDS round trips: 1830 clocks.)
template <int KIND>
__device__ __forceinline__ void bj_inner_sweep_k(unsigned char *smem, int2 IJ, int k, real thr, BigPsdCtl *ctl, int offslot, real *Qg, real *Sg,
                                                 int *flag_out) {
  real *S = reinterpret_cast<real *>(smem);
  real *Q = S + 2 * BJ_W * BJ_ILD; // (the second copy of S between them is the cross sweep's)
  RotCS *rot_cs = reinterpret_cast<RotCS *>(Q + BJ_W * BJ_ILD);
  __shared__ real red[BJ_UPD_THREADS_MAX / SCSAMD_WAVE];
  __shared__ int rot_any[2]; // (plain LDS accesses; a volatile flag is reached through FLAT loads / stores with system scope)
  __shared__ int rotated;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const bool worker = tid < BJ_INNER_THREADS;
  constexpr int nst = KIND == 0 ? BJ_W - 1 : (KIND == 1 ? BJ_B - 1 : BJ_B);
  for (int e = tid; e < BJ_W * BJ_W; e += nthr) {
    const int r = e & (BJ_W - 1), c = e >> 6;
    Q[r * BJ_ILD + c] = r == c ? (real)1 : (real)0;
  }
  if (tid == 0) rotated = 0;
  __syncthreads();
  real offmax = 0;
  constexpr int NP = BJ_B; // 32 pairs of the 64 local indices
  for (int st = 0; st < nst; ++st) {
    const int par = st & 1;
    if (tid < NP) {
      const int i = tid;
      const int2 pq_i = bj_inner_pair_k<KIND>(i, st);
      const int p = pq_i.x, q = pq_i.y;
      real c = 1, s = 0;
      const real apq = S[p * BJ_ILD + q], aqq = S[q * BJ_ILD + q], app = S[p * BJ_ILD + p];
      const real aa = absval(apq);
      const bool real_pair = bj_gidx(IJ, q) < k; // local order = global order (I < J): q is the larger index
      if (real_pair) offmax = aa > offmax ? aa : offmax;
      const bool rot = real_pair && aa > thr;
      if (rot) jacobi_cs(aqq - app, (real)2 * apq, c, s);
      rot_cs[i] = RotCS{c, s};
      const int vote = __any(rot ? 1 : 0); // the 32 lanes are half of one wave
      if (i == 0) rot_any[par] = vote;
    }
    __syncthreads();
    const int any = rot_any[par]; // uniform over the workgroup (this parity is written again two steps on, behind the next step's barrier)
    if (worker) {
      int i11[BJ_NBLK], i12[BJ_NBLK], i21[BJ_NBLK], i22[BJ_NBLK], ip[BJ_NROW], iq[BJ_NROW];
      RotCS r1[BJ_NBLK], r2[BJ_NBLK], rq[BJ_NROW];
      real a11[BJ_NBLK], a12[BJ_NBLK], a21[BJ_NBLK], a22[BJ_NBLK], vp[BJ_NROW], vq[BJ_NROW];
      bool same[BJ_NBLK];
#pragma unroll
      for (int u = 0; u < BJ_NBLK; ++u) {
        const int e = tid + u * BJ_INNER_THREADS, Qi = e / NP, P = e % NP;
        const int2 pq1 = bj_inner_pair_k<KIND>(P, st), pq2 = bj_inner_pair_k<KIND>(Qi, st);
        i11[u] = pq1.x * BJ_ILD + pq2.x;
        i12[u] = pq1.x * BJ_ILD + pq2.y;
        i21[u] = pq1.y * BJ_ILD + pq2.x;
        i22[u] = pq1.y * BJ_ILD + pq2.y;
        same[u] = P == Qi;
        r1[u] = rot_cs[P];
        r2[u] = rot_cs[Qi];
        a11[u] = S[i11[u]];
        a12[u] = S[i12[u]];
        a21[u] = S[i21[u]];
        a22[u] = S[i22[u]];
      }
#pragma unroll
      for (int j = 0; j < BJ_NROW; ++j) {
        const int f = tid + j * BJ_INNER_THREADS, Qi = f / BJ_W, i = f % BJ_W;
        const int2 pq2 = bj_inner_pair_k<KIND>(Qi, st);
        ip[j] = i * BJ_ILD + pq2.x;
        iq[j] = i * BJ_ILD + pq2.y;
        rq[j] = rot_cs[Qi];
        vp[j] = Q[ip[j]];
        vq[j] = Q[iq[j]];
      }
      if (any) { // a step in which no pair is above the threshold changes nothing
#pragma unroll
        for (int u = 0; u < BJ_NBLK; ++u) {
          const real c1 = r1[u].c, s1 = r1[u].s, c2 = r2[u].c, s2 = r2[u].s;
          const real r11 = c1 * a11[u] - s1 * a21[u], r12 = c1 * a12[u] - s1 * a22[u];
          const real r21 = s1 * a11[u] + c1 * a21[u], r22 = s1 * a12[u] + c1 * a22[u];
          const bool own = same[u] && s1 != (real)0; // the rotated pair's own off-diagonal entry: exact zero
          S[i11[u]] = c2 * r11 - s2 * r12;
          S[i12[u]] = own ? (real)0 : s2 * r11 + c2 * r12;
          S[i21[u]] = own ? (real)0 : c2 * r21 - s2 * r22;
          S[i22[u]] = s2 * r21 + c2 * r22;
        }
#pragma unroll
        for (int j = 0; j < BJ_NROW; ++j) {
          Q[ip[j]] = rq[j].c * vp[j] - rq[j].s * vq[j];
          Q[iq[j]] = rq[j].s * vp[j] + rq[j].c * vq[j];
        }
        if (tid == 0) rotated = 1;
      }
    }
    if (!any) continue; // nothing was written: the next step's rotations may be formed at once
    __syncthreads();
  }
  offmax = block_max(offmax, red); // (contains the barriers that make `rotated` and the last pass visible)
  if (tid == 0) {
    if (offmax > (real)0) atomicMax(&ctl->offmax_bits[offslot], bp_bits(offmax));
    *flag_out = rotated;
  }
  if (!rotated) return; // uniform: nobody reads Q / S' of a pair whose flag is 0
  for (int e = tid; e < BJ_W * BJ_W; e += nthr) {
    const int r = e & (BJ_W - 1), c = e >> 6;
    Qg[c * BJ_W + r] = Q[r * BJ_ILD + c];
    Sg[c * BJ_W + r] = S[r * BJ_ILD + c];
  }
}
// (measurement knobs of the cross sweep; the defaults are what ships)
#ifndef BJ_AHEAD_PRIO
#define BJ_AHEAD_PRIO 3
#endif
#ifndef BJ_AHEAD_ONE_BATCH
#define BJ_AHEAD_ONE_BATCH 1
#endif
#ifndef BJ_AHEAD_ANY_LATE
#define BJ_AHEAD_ANY_LATE 1
#endif
// one 2x2 block of J' S J: rows rotated by (c1, s1), then columns by (c2, s2); `own`: the block of a rotated pair with itself, whose
// off-diagonal entries are exact zeros.  ONE definition (explicit fused multiply-adds) for the lanes that update S and the lanes that
// look one step ahead, so that both get the same bits.
struct Blk2 {
  real n11, n12, n21, n22;
};
__device__ __forceinline__ Blk2 bj_block(real a11, real a12, real a21, real a22, RotCS r1, RotCS r2, bool own) {
  const real r11 = fma(r1.c, a11, -(r1.s * a21)), r12 = fma(r1.c, a12, -(r1.s * a22));
  const real r21 = fma(r1.s, a11, r1.c * a21), r22 = fma(r1.s, a12, r1.c * a22);
  Blk2 o;
  o.n11 = fma(r2.c, r11, -(r2.s * r12));
  o.n12 = own ? (real)0 : fma(r2.s, r11, r2.c * r12);
  o.n21 = own ? (real)0 : fma(r2.c, r21, -(r2.s * r22));
  o.n22 = fma(r2.s, r21, r2.c * r22);
  return o;
}

// The CROSS sweep (pairs p in I, q in J; 32 steps; 31 of the 32 launches of a sweep of a 1024 x 1024 block).  Three things set it apart
// from the generic sweep above (measurements: profiles/r4_psd_fused_step.md):
//  * software pipeline: the rotations of step st + 1 are formed DURING the update of step st by a wave that does nothing else (the
//    two-phase step is 910 clocks of rotations -- an LDS round trip, the rsqrt chain, a barrier -- plus 2020 of update).  S is double-
//    buffered (the update reads one copy and writes the other), so the look-ahead lanes can read the old entries while they are being
//    replaced and apply the current step's rotations to just the three entries their next pair needs (S[p'][p'], S[q'][q'], S[p'][q']:
//    three 2x2 blocks, the same bj_block as the update, hence the same bits).  One barrier per step.  Round 3 measured this form slower
//    (114 vs 96 us per sweep): the look-ahead chain then went through pair tables and an fp64 sqrt / div / rsqrt sequence.
//  * the update moves and computes less: only the UPPER triangle of the symmetric S is kept (r <= c in local order; 496 off-diagonal
//    2x2 blocks P < Q and 32 diagonal ones instead of 1024 blocks), and the I column of every (row, pair) item of Q never leaves its
//    lane's registers (a lane keeps its items for the whole sweep; only the J columns, which change partner every step, live in LDS):
//    5280 + 6144 LDS accesses per step instead of 20480, two thirds of the arithmetic.  (It bought 1 %: a step turned out to be a
//    chain of latencies -- barrier, the flag's and the operands' LDS round trips, ~130 instructions per worker wave, the stores'
//    completion: 0.98 us -- not a throughput problem.  Kept because it is the leaner form.)
//  * waves 0-7 carry the update (a block of S and four items of Q per lane; the arithmetic is a quarter of a wave's instructions, so
//    fewer waves with more items each beat fifteen waves with one or two), wave 15 looks ahead, the others only keep the barriers company.
__device__ __forceinline__ void bj_inner_sweep_cross(unsigned char *smem, int2 IJ, int k, real thr, BigPsdCtl *ctl, int offslot, real *Qg, real *Sg,
                                                     int *flag_out) {
  real *S0 = reinterpret_cast<real *>(smem);
  real *Q = S0 + 2 * BJ_W * BJ_ILD;                              // only its J columns (32 .. 63) are used here
  RotCS *rot_cs = reinterpret_cast<RotCS *>(Q + BJ_W * BJ_ILD); // [2][BJ_B]: generation st & 1 holds the rotations of step st
  __shared__ real red[BJ_UPD_THREADS_MAX / SCSAMD_WAVE];
  __shared__ int rot_any[2]; // generation st & 1: does step st rotate anything?
  __shared__ int rotated;
  const int tid = threadIdx.x, nthr = blockDim.x; // nthr == BJ_UPD_THREADS_MAX
  constexpr int nst = BJ_B;
  constexpr int AHEAD0 = BJ_UPD_THREADS_MAX - SCSAMD_WAVE, NOFF = BJ_B * (BJ_B - 1) / 2;
  const bool ahead = tid >= AHEAD0;       // the wave that forms the rotations (its two halves do the same work: no divergence; the lanes vote)
  const int ai = (tid - AHEAD0) & (BJ_B - 1), aj = (ai + 1) & (BJ_B - 1);
  // ---- this lane's items, fixed for the sweep.  Waves 0-7: one block of S (lanes 0 .. 495 the off-diagonal blocks P < Q, lanes 496 .. 511
  // the diagonal blocks 0 .. 15) and four (row, pair) items of Q; wave 8, lanes 0 .. 15: the diagonal blocks 16 .. 31.
  constexpr int NQ = 4, QSTEP = 8;
  const bool qworker = tid < BJ_INNER_THREADS;
  const int qrow = tid & (BJ_W - 1), qpair0 = (tid >> 6) & 7; // items of Q: row qrow, pairs qpair0 + 8 j  (the items on four waves of their
                                                              // own, eight per lane, the S workers carrying only their block: 12.4 vs 11.1 ms)
  int sP = -1, sQ = -1; // block of S: rows {sP, q(sP)} x columns {sQ, q(sQ)}, sP <= sQ (q(i) = 32 + (i + st) mod 32)
  if (tid < NOFF) {     // t -> (P < Q): column Q of the strict upper triangle holds Q blocks
    int Qc = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)tid)) * 0.5f);
    while (Qc * (Qc - 1) / 2 > tid) --Qc;
    while ((Qc + 1) * Qc / 2 <= tid) ++Qc;
    sQ = Qc;
    sP = tid - Qc * (Qc - 1) / 2;
  } else if (tid < NOFF + BJ_B) {
    sP = sQ = tid - NOFF;
  }
  const bool sworker = sP >= 0, diag = sP == sQ;
  real qI[NQ]; // Q[row][pair's I column]: stays here
#pragma unroll
  for (int j = 0; j < NQ; ++j) qI[j] = qrow == qpair0 + QSTEP * j ? (real)1 : (real)0;
  for (int e = tid; e < BJ_W * BJ_B; e += nthr) { // J columns of Q <- those of the identity
    const int r = e & (BJ_W - 1), c = BJ_B + (e >> 6);
    Q[r * BJ_ILD + c] = r == c ? (real)1 : (real)0;
  }
  if (tid == 0) rotated = 0;
  __syncthreads();
  real offmax = 0;
  // the rotation of pair (ai, q) from its three entries, recorded as generation `gen`; the 32 lanes vote on "anything to rotate"
#define BJ_FORM(gen_, q_, app_, aqq_, apq_)                                                                 \
  do {                                                                                                      \
    real c_ = 1, s_ = 0;                                                                                    \
    const real apq__ = (apq_), aa_ = absval(apq__);                                                         \
    const bool real_pair_ = bj_gidx(IJ, (q_)) < k; /* local order = global order (I < J): q is the larger */ \
    if (real_pair_) offmax = aa_ > offmax ? aa_ : offmax;                                                   \
    const bool rot_ = real_pair_ && aa_ > thr;                                                              \
    if (rot_) jacobi_cs((aqq_) - (app_), (real)2 * apq__, c_, s_);                                          \
    rot_cs[(gen_) * BJ_B + ai] = RotCS{c_, s_};                                                             \
    const int vote_ = __any(rot_ ? 1 : 0);                                                                  \
    if (ai == 0) rot_any[(gen_)] = vote_;                                                                   \
  } while (0)
  // offsets of the ten entries the look-ahead lane reads in step st.  Next step's pair is p' = ai, q' = qn = 32 + (ai + st + 1) mod 32;
  // in step st p' is the first index of pair ai (partner qa) and q' the second index of pair aj = (ai + 1) mod 32.
  //   diagonal block (ai, ai): its n11 is S'[p'][p'];  diagonal block (aj, aj): its n22 is S'[q'][q'];
  //   block (min, max of ai, aj): its n12 (ai < aj) or n21 (ai = 31, aj = 0) is S'[p'][q']
  int ao[10];
  const int fP = ai < aj ? ai : aj, fQ = ai < aj ? aj : ai;
  auto ahead_offsets = [&](int st) {
    const int qa = BJ_B + ((ai + st) & (BJ_B - 1)), qn = BJ_B + ((ai + st + 1) & (BJ_B - 1));
    const int qP = ai < aj ? qa : qn, qQ = ai < aj ? qn : qa, lo = qP < qQ ? qP : qQ, hi = qP < qQ ? qQ : qP;
    ao[0] = ai * BJ_ILD + ai, ao[1] = ai * BJ_ILD + qa, ao[2] = qa * BJ_ILD + qa;
    ao[3] = aj * BJ_ILD + aj, ao[4] = aj * BJ_ILD + qn, ao[5] = qn * BJ_ILD + qn;
    ao[6] = fP * BJ_ILD + fQ, ao[7] = fP * BJ_ILD + qQ, ao[8] = fQ * BJ_ILD + qP, ao[9] = lo * BJ_ILD + hi;
  };
  if (ahead) { // step 0's rotations from the matrix as loaded
    const int p = ai, q = BJ_B + ai;
    BJ_FORM(0, q, S0[p * BJ_ILD + p], S0[q * BJ_ILD + q], S0[p * BJ_ILD + q]);
    ahead_offsets(0);
  }
  __syncthreads();
  if (BJ_AHEAD_PRIO && ahead) __builtin_amdgcn_s_setprio(BJ_AHEAD_PRIO);
  int cur = 0;
  for (int st = 0; st < nst; ++st) {
    const int gen = st & 1;
    const real *Sr = S0 + cur * BJ_W * BJ_ILD;
    real *Sw = S0 + (cur ^ 1) * BJ_W * BJ_ILD;
    const RotCS *cs = rot_cs + gen * BJ_B;
    int any; // does this step rotate anything?  Uniform (set before the barrier that ended the previous step).  The look-ahead wave asks
             // for it BEHIND its operand reads: those are the step's first LDS requests
    if (!ahead) {
      any = rot_any[gen];
      if (any) { // a step in which no pair is above the threshold changes nothing
        real vq[NQ];
        RotCS rq[NQ];
        int oq[NQ];
        if (qworker) {
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            const int Qi = qpair0 + QSTEP * j;
            oq[j] = qrow * BJ_ILD + BJ_B + ((Qi + st) & (BJ_B - 1));
            rq[j] = cs[Qi];
            vq[j] = Q[oq[j]];
          }
        }
        if (sworker) { // every read of the lane in one batch (the items of Q above included), then the arithmetic, then the stores
          const int qP = BJ_B + ((sP + st) & (BJ_B - 1)), qQ = BJ_B + ((sQ + st) & (BJ_B - 1));
          const int lo = qP < qQ ? qP : qQ, hi = qP < qQ ? qQ : qP;
          const int o11 = sP * BJ_ILD + sQ, o12 = sP * BJ_ILD + qQ, o21 = sQ * BJ_ILD + qP, o22 = lo * BJ_ILD + hi; // (diagonal block: o21 == o12)
          const RotCS r1 = cs[sP], r2 = cs[sQ];
          const real a11 = Sr[o11], a12 = Sr[o12], a21 = Sr[o21], a22 = Sr[o22];
          const Blk2 o = bj_block(a11, a12, a21, a22, r1, r2, diag && r1.s != (real)0);
          Sw[o11] = o.n11;
          Sw[o12] = o.n12;
          Sw[o21] = diag ? o.n12 : o.n21; // (diagonal block: the same address as o12)
          Sw[o22] = o.n22;
        }
        if (qworker) {
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            const real vp = qI[j];
            qI[j] = fma(rq[j].c, vp, -(rq[j].s * vq[j]));
            Q[oq[j]] = fma(rq[j].s, vp, rq[j].c * vq[j]);
          }
        }
        if (tid == 0) rotated = 1;
      }
    } else if (st + 1 < nst) {
      // the ten offsets were formed before the barrier: these reads are the wave's first instructions of the step
      const real d11 = Sr[ao[0]], d12 = Sr[ao[1]], d22 = Sr[ao[2]];
      const real e11 = Sr[ao[3]], e12 = Sr[ao[4]], e22 = Sr[ao[5]];
      const real f11 = Sr[ao[6]], f12 = Sr[ao[7]], f21 = Sr[ao[8]], f22 = Sr[ao[9]];
      const RotCS ri = cs[ai], rj = cs[aj];
#if BJ_AHEAD_ANY_LATE
      any = rot_any[gen];
#endif
#if BJ_AHEAD_ONE_BATCH
      __builtin_amdgcn_sched_barrier(0); // all reads in ONE round trip (left to itself the scheduler split them into three)
#endif
      const int qn = BJ_B + ((ai + st + 1) & (BJ_B - 1));
      const Blk2 bd = bj_block(d11, d12, d12, d22, ri, ri, ri.s != (real)0);
      const Blk2 be = bj_block(e11, e12, e12, e22, rj, rj, rj.s != (real)0);
      const Blk2 bf = bj_block(f11, f12, f21, f22, ai < aj ? ri : rj, ai < aj ? rj : ri, false);
      BJ_FORM(gen ^ 1, qn, bd.n11, be.n22, ai < aj ? bf.n12 : bf.n21);
      ahead_offsets(st + 1);
#if !BJ_AHEAD_ANY_LATE
      any = rot_any[gen];
#endif
    } else {
      any = rot_any[gen];
    }
    __syncthreads();
    if (any) cur ^= 1;
  }
#undef BJ_FORM
  offmax = block_max(offmax, red); // (contains the barriers that make `rotated` visible)
  if (tid == 0) {
    if (offmax > (real)0) atomicMax(&ctl->offmax_bits[offslot], bp_bits(offmax));
    *flag_out = rotated;
  }
  if (!rotated) return; // uniform: nobody reads Q / S' of a pair whose flag is 0
  const real *Sf = S0 + cur * BJ_W * BJ_ILD;
  if (qworker) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) Qg[(qpair0 + QSTEP * j) * BJ_W + qrow] = qI[j]; // I columns: from the registers
  }
  for (int e = tid; e < BJ_W * BJ_B; e += nthr) {                            // J columns: from LDS
    const int r = e & (BJ_W - 1), c = BJ_B + (e >> 6);
    Qg[c * BJ_W + r] = Q[r * BJ_ILD + c];
  }
  for (int e = tid; e < BJ_W * BJ_W; e += nthr) { // S' from the triangle that was kept
    const int r = e & (BJ_W - 1), c = e >> 6;
    Sg[c * BJ_W + r] = r <= c ? Sf[r * BJ_ILD + c] : Sf[c * BJ_ILD + r];
  }
}

__device__ __forceinline__ void bj_inner_sweep(unsigned char *smem, int2 IJ, int kind, int k, real thr, BigPsdCtl *ctl, int offslot, real *Qg,
                                               real *Sg, int *flag_out) {
  if (kind == 2) bj_inner_sweep_cross(smem, IJ, k, thr, ctl, offslot, Qg, Sg, flag_out);
  else if (kind == 1) bj_inner_sweep_k<1>(smem, IJ, k, thr, ctl, offslot, Qg, Sg, flag_out);
  else bj_inner_sweep_k<0>(smem, IJ, k, thr, ctl, offslot, Qg, Sg, flag_out);
}

// Qbuf / Sbuf: per (block, pair) 64 x 64 column-major; Qflag: 1 if the sweep rotated anything
__global__ __launch_bounds__(BJ_INNER_LAUNCH) void k_bj_inner(BigPsdView B, real *Qbuf, real *Sbuf, int *Qflag, int npmax, int arg, int cross) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bj_smem[];
  real *S = reinterpret_cast<real *>(bj_smem);
  const int slot = arg & 1, step = arg >> 1;
  const int b = blockIdx.y, pi = blockIdx.x, tid = threadIdx.x;
  BigPsdCtl *ctl = B.ctl + b;
  const int done = ctl->done, kraw = ctl->kraw, cur = ctl->cur[slot];
  const real thr = ctl->thr;
  const BlockShape sh = bp_shape_raw(kraw);
  if (done || step >= bj_outer_steps(sh.nbc, cross) || pi >= sh.nbc / 2) return;
  const int2 IJ = bj_pair_sched(pi, step, sh.nbc, cross);
  const int kind = !cross ? 0 : (step == 0 ? 1 : 2);
  const size_t ld = B.ld, mat = (size_t)b * ld * ld;
  const real *Aold = (cur ? B.A1 : B.A) + mat;
  for (int e = tid; e < BJ_W * BJ_W; e += BJ_INNER_LAUNCH) {
    const int r = e & (BJ_W - 1), c = e >> 6; // r fast: 32-entry runs of a column of A
    S[r * BJ_ILD + c] = Aold[(size_t)bj_gidx(IJ, c) * ld + bj_gidx(IJ, r)];
  }
  const size_t slot_q = ((size_t)b * npmax + pi);
  bj_inner_sweep(bj_smem, IJ, kind, sh.k, thr, ctl, ctl->sweeps & 1, Qbuf + slot_q * BJ_W * BJ_W, Sbuf + slot_q * BJ_W * BJ_W, Qflag + slot_q);
}

#ifndef BJ_UPD_THREADS_OVERRIDE
constexpr int BJ_UPD_THREADS = 1024; // sixteen waves: one 16 x 16 output tile each per product (measured 1024 x 1024: 26.1 / 24.6 / 23.8 ms at 256 / 512 / 1024)
#else
constexpr int BJ_UPD_THREADS = BJ_UPD_THREADS_OVERRIDE;
#endif
// O[i][j] = sum_k L[i][k] R[k][j] for operands in LDS, 64 summation indices, i < 16 mt, j < 16 nt: L[i][k] at L[i * BJ_LD + k], R[k][j] at
// R[j * BJ_LD + k] (both contiguous along the summation index), O[i][j] at O[j * BJ_LD + i].  16 x 16 output tiles dealt to the waves;
// lane (li = l & 15, lk = l >> 4) supplies L[16 ti + li][4 ks + lk] and R[4 ks + lk][16 tj + li].  An output entry is the same sequence of
// operations whatever mt, nt and the number of waves are (the fused step computes a quarter of a tile product and must get the
// update's bits).
__device__ __forceinline__ void bj_gemm_part(const real *L, const real *R, real *O, int mt, int nt, int tid, int nthr) {
#ifndef SFLOAT
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  for (int t = wave; t < mt * nt; t += nthr / SCSAMD_WAVE) {
    const int ti = t % mt, tj = t / mt;
    const real *lp = L + (ti * 16 + li) * BJ_LD + lk, *rp = R + (tj * 16 + li) * BJ_LD + lk;
    f64x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < BJ_W / 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lp[ks * 4], rp[ks * 4], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) O[(tj * 16 + li) * BJ_LD + ti * 16 + lk + 4 * r] = acc[r];
  }
#else
  for (int e = tid; e < 256 * mt * nt; e += nthr) {
    const int i = e % (16 * mt), j = e / (16 * mt);
    real acc = 0;
    for (int kk = 0; kk < BJ_W; ++kk) acc += L[i * BJ_LD + kk] * R[j * BJ_LD + kk];
    O[j * BJ_LD + i] = acc;
  }
#endif
}
__device__ __forceinline__ void bj_gemm64(const real *L, const real *R, real *O, int tid) { bj_gemm_part(L, R, O, 4, 4, tid, BJ_UPD_THREADS); }

// one tile of the step `step` of block b: tile < nta -> A tile (P <= Q) and its mirror, else a V tile.  smem: 4 x 64 x BJ_LD reals.
// left_bits (round 6; the LAST step of the block's sweep only, else null): the largest off-diagonal entry this job writes goes there --
// the sweep's closing reads off it whether another sweep would rotate anything (see k_bp_offscan, which does the same as a pass of its
// own for the forms that do not come through here), at the price of one atomic per workgroup instead of a launch per sweep.
__device__ __forceinline__ void bj_update_job(const BigPsdView &B, unsigned char *smem, const real *__restrict__ Qbuf, const real *__restrict__ Sbuf,
                                              const int *__restrict__ Qflag, int npmax, int b, int tile, int step, int cross, const BlockShape &sh,
                                              const real *Aold, real *Anew, real *V, unsigned long long *left_bits = nullptr) {
  real *xs = reinterpret_cast<real *>(smem);
  real *qp = xs + BJ_W * BJ_LD, *qq = qp + BJ_W * BJ_LD, *ts = qq + BJ_W * BJ_LD;
  __shared__ real red_left[BJ_UPD_THREADS_MAX / SCSAMD_WAVE];
  const int tid = threadIdx.x;
  const size_t ld = B.ld;
  // Only the tiles (P, Q) with P <= Q are computed; the mirror tile (Q, P) receives the transpose, so A stays EXACTLY symmetric
  // from step to step (two independent products Q_P' A[P,Q] Q_Q and Q_Q' A[Q,P] Q_P sum in different orders: symmetric only to
  // rounding, and k_bj_inner builds its rotations from one triangle) and the step does half the matrix-core work.
  const int np = sh.nbc / 2, nta = np * (np + 1) / 2, TR = sh.K64 / BJ_W;
  if (tile >= nta + np * TR) return;
  const int *flags = Qflag + (size_t)b * npmax;
  const real *Qb = Qbuf + (size_t)b * npmax * BJ_W * BJ_W, *Sb = Sbuf + (size_t)b * npmax * BJ_W * BJ_W;
  if (tile < nta) {
    // tile -> (Pp <= Qp): column Qp of the upper triangle holds Qp + 1 tiles
    int Qp = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
    while (Qp * (Qp + 1) / 2 > tile) --Qp;
    while ((Qp + 1) * (Qp + 2) / 2 <= tile) ++Qp;
    const int Pp = tile - Qp * (Qp + 1) / 2;
    const int2 IJp = bj_pair_sched(Pp, step, sh.nbc, cross), IJq = bj_pair_sched(Qp, step, sh.nbc, cross);
    const int fP = flags[Pp], fQ = flags[Qp];
    real left = 0; // largest |entry| written at (row, column) with row != column, both below k (the mirror entry is the same number)
    auto seen = [&](int gi, int gj, real v) {
      const real a = absval(v);
      if (gi != gj && gi < sh.k && gj < sh.k && a > left) left = a;
    };
    if (Pp == Qp) {
      if (fP) { // the pair's own tile: the inner sweep's S' (exact zeros, relatively accurate small entries)
        for (int e = tid; e < BJ_W * BJ_W; e += BJ_UPD_THREADS) {
          const int i = e & (BJ_W - 1), j = e >> 6, gi = bj_gidx(IJp, i), gj = bj_gidx(IJq, j);
          const real v = Sb[(size_t)Pp * BJ_W * BJ_W + j * BJ_W + i];
          Anew[(size_t)gj * ld + gi] = v;
          seen(gi, gj, v);
        }
      } else {
        for (int e = tid; e < BJ_W * BJ_W; e += BJ_UPD_THREADS) {
          const int i = e & (BJ_W - 1), j = e >> 6, gi = bj_gidx(IJp, i), gj = bj_gidx(IJq, j);
          const size_t g = (size_t)gj * ld + gi;
          const real v = Aold[g];
          Anew[g] = v;
          seen(gi, gj, v);
        }
      }
    } else if (!fP && !fQ) { // neither pair rotated: the tile and its mirror move to the other copy as they are
      for (int e = tid; e < BJ_W * BJ_W; e += BJ_UPD_THREADS) {
        const int i = e & (BJ_W - 1), j = e >> 6, gi = bj_gidx(IJp, i), gj = bj_gidx(IJq, j);
        const size_t g = (size_t)gj * ld + gi, gm = (size_t)bj_gidx(IJp, j) * ld + bj_gidx(IJq, i);
        const real v = Aold[g], vm = Aold[gm];
        Anew[g] = v;
        Anew[gm] = vm;
        seen(gi, gj, v);
        seen(bj_gidx(IJq, i), bj_gidx(IJp, j), vm);
      }
    } else {
      // X[i][k] = A[P_i, Q_k]; Q_Q[k][j] and Q_P[k][i], identity where the pair did not rotate
#pragma unroll 4
      for (int e = tid; e < BJ_W * BJ_W; e += BJ_UPD_THREADS) {
        const int i = e & (BJ_W - 1), kk = e >> 6;
        xs[i * BJ_LD + kk] = Aold[(size_t)bj_gidx(IJq, kk) * ld + bj_gidx(IJp, i)];
        // Qbuf is column-major: Q[k][j] at j * 64 + k -> qq[j * BJ_LD + k]: here (kk, i) play (j, k)
        qq[kk * BJ_LD + i] = fQ ? Qb[(size_t)Qp * BJ_W * BJ_W + kk * BJ_W + i] : (i == kk ? (real)1 : (real)0);
        qp[kk * BJ_LD + i] = fP ? Qb[(size_t)Pp * BJ_W * BJ_W + kk * BJ_W + i] : (i == kk ? (real)1 : (real)0);
      }
      __syncthreads();
      bj_gemm64(xs, qq, ts, tid); // T = X Q_Q, T[i][j] at ts[j * BJ_LD + i]
      __syncthreads();
      bj_gemm64(qp, ts, xs, tid); // O = Q_P' T: L[i][k] = Q_P[k][i] = qp[i * BJ_LD + k], R[k][j] = T[k][j] = ts[j * BJ_LD + k]
      __syncthreads();
      for (int e = tid; e < BJ_W * BJ_W; e += BJ_UPD_THREADS) {
        const int i = e & (BJ_W - 1), j = e >> 6, gi = bj_gidx(IJp, i), gj = bj_gidx(IJq, j);
        const real v = xs[j * BJ_LD + i];
        Anew[(size_t)gj * ld + gi] = v;                                                   // A'[P_i, Q_j]
        Anew[(size_t)bj_gidx(IJp, j) * ld + bj_gidx(IJq, i)] = xs[i * BJ_LD + j];         // A'[Q_i, P_j] = A'[P_j, Q_i]
        seen(gi, gj, v);
      }
    }
    if (left_bits) { // (uniform)
      left = block_max(left, red_left);
      // (the relaxed read may be behind the other workgroups' maxima, i.e. too small: a needless atomic then, never a missing one)
      if (tid == 0 && left > bp_from_bits(__hip_atomic_load(left_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicMax(left_bits, bp_bits(left));
    }
    return;
  }
  // ---- V[:, Q] <- V[:, Q] Q_Q on a tile of 64 rows (in place: the workgroup owns the entries it rewrites)
  tile -= nta;
  const int Qp = tile / TR, r0 = (tile % TR) * BJ_W;
  if (!flags[Qp]) return;
  const int2 IJq = bj_pair_sched(Qp, step, sh.nbc, cross);
#pragma unroll 4
  for (int e = tid; e < BJ_W * BJ_W; e += BJ_UPD_THREADS) {
    const int i = e & (BJ_W - 1), kk = e >> 6;
    xs[i * BJ_LD + kk] = V[(size_t)bj_gidx(IJq, kk) * ld + r0 + i];
    qq[kk * BJ_LD + i] = Qb[(size_t)Qp * BJ_W * BJ_W + kk * BJ_W + i];
  }
  __syncthreads();
  bj_gemm64(xs, qq, ts, tid);
  __syncthreads();
  for (int e = tid; e < BJ_W * BJ_W; e += BJ_UPD_THREADS) {
    const int i = e & (BJ_W - 1), j = e >> 6;
    V[(size_t)bj_gidx(IJq, j) * ld + r0 + i] = ts[j * BJ_LD + i];
  }
}


__global__ __launch_bounds__(BJ_UPD_THREADS) void k_bj_update(BigPsdView B, const real *__restrict__ Qbuf, const real *__restrict__ Sbuf,
                                                          const int *__restrict__ Qflag, int npmax, int arg, int cross) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bj_smem[];
  const int slot = arg & 1, step = arg >> 1;
  const int b = blockIdx.y, tid = threadIdx.x;
  BigPsdCtl *ctl = B.ctl + b;
  const int done = ctl->done, kraw = ctl->kraw, cur = ctl->cur[slot];
  const BlockShape sh = bp_shape_raw(kraw);
  const bool active = !done && step < bj_outer_steps(sh.nbc, cross);
  if (blockIdx.x == 0 && tid == 0) ctl->cur[slot ^ 1] = active ? cur ^ 1 : cur;
  if (!active) return;
  const size_t ld = B.ld, mat = (size_t)b * ld * ld;
  bj_update_job(B, bj_smem, Qbuf, Sbuf, Qflag, npmax, b, blockIdx.x, step, cross, sh, (cur ? B.A1 : B.A) + mat, (cur ? B.A : B.A1) + mat, B.V + mat);
}

// ---- the fused step (round 4; VERDICT r3 item 2: "overlap outer step k+1's k_bj_inner with step k's k_bj_update").
// The inner sweep is one workgroup per pair of block columns (16 workgroups for a 1024 x 1024 block) and a chain of 32 dependent
// steps: 40 us in which the rest of the chip idles; the update is 14 us of the whole chip.  As two launches they add up.  This launch
// does both: workgroups [0, npmax) run the inner sweep of step k + 1, the others the update of step k.  The inner sweep of step k + 1
// needs A as the update of step k leaves it -- only the 64 x 64 diagonal subproblem of ITS pair (I', J'), which it forms itself from
// what the update reads (the matrix before step k, Q and S' of step k's pairs): the quadrants [I', I'] and [J', J'] are pieces of the
// diagonal tiles of step k (S' of the pair the block column was in), the quadrant [I', J'] is a 32 x 32 piece of
// Q_P' A[P, Q] Q_Q for the two step-k pairs P, Q that held I' and J' -- the same tile product, in the same orientation (P <= Q) and
// operation order as bj_update_job computes it, so the subproblem has the bits the update writes to the other copy of A.
// Q / S' / flags are double-buffered by launch (the update of step k reads step k's while the inner sweep writes step k + 1's).
// The last launch of a sweep runs the FIRST inner sweep of the next sweep before k_bp_sweep_end has closed this one (its
// off-diagonal maximum goes to the other parity slot of the control record); if the block turns out converged the work is dropped.
static_assert(BJ_UPD_THREADS == BJ_UPD_THREADS_MAX && BJ_INNER_LAUNCH == BJ_UPD_THREADS_MAX, "bj_inner_sweep_cross lays its roles out over a 1024-thread workgroup");
struct BjFusedArgs {
  int slot;        // launch parity of the control record's `cur` (as `arg & 1` of the two-launch form)
  int do_update;   // 0: the first launch of a projection (inner sweep of step 0 only)
  int upd_step;    // outer step the update workgroups apply
  int inn_step;    // outer step of the inner sweep
  int next_sweep;  // inner sweep belongs to the sweep after the current one (inn_step == 0 of it)
  int qin, qout;   // which of the two Q / S' / flag buffers the update reads / the inner sweep writes
  int cross;
  int scan;        // the last update of a block's sweep records the largest off-diagonal entry it writes (no k_bp_offscan launch)
  int flat;        // inner sweep's prologue with ONE level of global loads (0: rounds 4-5's three levels; A/B measurements)
  int one_d;       // one-dimensional grid, inner-sweep workgroups of all blocks first (0: x = job, y = block; A/B measurements)
};

__global__ __launch_bounds__(BJ_UPD_THREADS) void k_bj_fused(BigPsdView B, real *Qbuf2, real *Sbuf2, int *Qflag2, int npmax, BjFusedArgs F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bj_smem[];
  // One-dimensional grid (round 6): the inner-sweep workgroups of ALL blocks first, then the update workgroups.  Workgroups are handed
  // out in the order of their index and one fits a CU (135 KB of LDS); with (x = job, y = block) the inner sweeps of the later blocks
  // -- the launch's critical path, ~40 us -- queued behind the ~11 us update jobs of the earlier ones whenever a launch has more
  // workgroups than the chip has CUs (16 blocks of order 200: 480; 4 of 512: 432).  bx: the job index of the old x dimension.
  const int n_inner = B.nbig * npmax, tid = threadIdx.x;
  int b, bx;
  if (!F.one_d) {
    b = blockIdx.y;
    bx = blockIdx.x;
  } else if ((int)blockIdx.x < n_inner) {
    b = (int)blockIdx.x / npmax;
    bx = (int)blockIdx.x - b * npmax;
  } else {
    const int u = (int)blockIdx.x - n_inner, per = ((int)gridDim.x - n_inner) / B.nbig;
    b = u / per;
    bx = npmax + (u - b * per);
  }
  BigPsdCtl *ctl = B.ctl + b;
  const int done = ctl->done, kraw = ctl->kraw, cur = ctl->cur[F.slot];
  const BlockShape sh = bp_shape_raw(kraw);
  const int ost = bj_outer_steps(sh.nbc, F.cross);
  const bool upd_active = F.do_update && !done && F.upd_step < ost;
  const size_t ld = B.ld, mat = (size_t)b * ld * ld;
  const real *Aold = (cur ? B.A1 : B.A) + mat;
  const size_t qsz = (size_t)B.nbig * npmax * BJ_W * BJ_W, fsz = (size_t)B.nbig * npmax;
  const real *Qin = Qbuf2 + F.qin * qsz, *Sin = Sbuf2 + F.qin * qsz;
  const int *Fin = Qflag2 + F.qin * fsz;
  if (bx >= npmax) { // ---- update of step upd_step
    if (!F.do_update) return;
    if (bx == npmax && tid == 0) ctl->cur[F.slot ^ 1] = upd_active ? cur ^ 1 : cur;
    if (!upd_active) return;
    bj_update_job(B, bj_smem, Qin, Sin, Fin, npmax, b, bx - npmax, F.upd_step, F.cross, sh, Aold, (cur ? B.A : B.A1) + mat, B.V + mat,
                  F.scan && F.upd_step == ost - 1 ? &ctl->left_bits : nullptr); // the last update of THIS block's sweep looks at what it leaves
    return;
  }
  // ---- inner sweep of step inn_step on the matrix as the update of this launch leaves it
  const int pi = bx;
  if (done || pi >= sh.nbc / 2 || (!F.next_sweep && F.inn_step >= ost)) return;
  const real thr = ctl->thr;
  const int2 IJ = bj_pair_sched(pi, F.inn_step, sh.nbc, F.cross);
  const int kind = !F.cross ? 0 : (F.inn_step == 0 ? 1 : 2);
  real *S = reinterpret_cast<real *>(bj_smem);
  const int nthr = BJ_UPD_THREADS;
  if (!upd_active) { // (a smaller block whose sweep is over, or the first launch) the matrix is not changing under this launch
    for (int e = tid; e < BJ_W * BJ_W; e += nthr) {
      const int r = e & (BJ_W - 1), c = e >> 6;
      S[r * BJ_ILD + c] = Aold[(size_t)bj_gidx(IJ, c) * ld + bj_gidx(IJ, r)];
    }
  } else {
    __shared__ int s_prev[4]; // pair and half (0: it was the pair's I, 1: its J) of I' and of J' in step upd_step
    const int np = sh.nbc / 2;
    if (tid < np) {
      const int2 pr = bj_pair_sched(tid, F.upd_step, sh.nbc, F.cross);
      if (pr.x == IJ.x || pr.y == IJ.x) { s_prev[0] = tid; s_prev[1] = pr.y == IJ.x; }
      if (pr.x == IJ.y || pr.y == IJ.y) { s_prev[2] = tid; s_prev[3] = pr.y == IJ.y; }
    }
    __syncthreads();
    const int PI = s_prev[0], hI = s_prev[1], PJ = s_prev[2], hJ = s_prev[3];
    const int *flags = Fin + (size_t)b * npmax;
    const real *Qb = Qin + (size_t)b * npmax * BJ_W * BJ_W, *Sb = Sin + (size_t)b * npmax * BJ_W * BJ_W;
    const int2 prI = bj_pair_sched(PI, F.upd_step, sh.nbc, F.cross), prJ = bj_pair_sched(PJ, F.upd_step, sh.nbc, F.cross);
    if (F.flat) {
      // Round 6: ONE level of global loads.  Everything the subproblem is made of is asked for at once -- both candidates of every entry
      // (S' / Q of the pairs of step k, valid memory whether or not they rotated, and A itself) beside the two flags that pick between
      // them -- and selected when it is all here.  (Round 4's form below: flags, then the diagonal quadrants, then the tile and the Q
      // columns, each level waiting for the one before -- three far round trips on the critical path of every launch.)
      static_assert(BJ_UPD_THREADS == BJ_B * BJ_B, "flat prologue: one entry per quadrant and lane");
      const int x = tid & (BJ_B - 1), y = tid >> 5;
      const int fI = flags[PI], fJ = flags[PJ];
      const size_t tile = (size_t)BJ_W * BJ_W;
      auto sprime = [&](int P, int rr, int cc) -> real { return Sb[(size_t)P * tile + cc * BJ_W + rr]; };
      auto aentry = [&](int2 pr, int rr, int cc) -> real { return Aold[(size_t)bj_gidx(pr, cc) * ld + bj_gidx(pr, rr)]; };
      const real dIs = sprime(PI, hI * BJ_B + x, hI * BJ_B + y), dIa = aentry(prI, hI * BJ_B + x, hI * BJ_B + y);
      const real dJs = sprime(PJ, hJ * BJ_B + x, hJ * BJ_B + y), dJa = aentry(prJ, hJ * BJ_B + x, hJ * BJ_B + y);
      const bool swap = PI > PJ;
      const int Pa = swap ? PJ : PI, Pb = swap ? PI : PJ, ha = swap ? hJ : hI, hb = swap ? hI : hJ;
      const int2 pra = swap ? prJ : prI, prb = swap ? prI : prJ;
      real cs_ = 0, ca_ = 0, xv[4] = {0, 0, 0, 0}, qv[2] = {0, 0}, pv[2] = {0, 0};
      if (PI == PJ) { // (uniform) I' and J' sat in ONE pair of step k: the quadrant is a piece of that pair's diagonal tile too
        cs_ = sprime(PI, hI * BJ_B + x, hJ * BJ_B + y);
        ca_ = aentry(prI, hI * BJ_B + x, hJ * BJ_B + y);
      } else {
        ca_ = Aold[(size_t)bj_gidx(IJ, BJ_B + y) * ld + bj_gidx(IJ, x)]; // neither pair rotated: the update copies the tile
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int e = tid + t * BJ_UPD_THREADS, i = e & (BJ_W - 1), kk = e >> 6;
          xv[t] = Aold[(size_t)bj_gidx(prb, kk) * ld + bj_gidx(pra, i)];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int e = tid + t * BJ_UPD_THREADS, i = e & (BJ_W - 1), j = e >> 6;
          qv[t] = Qb[(size_t)Pb * tile + (hb * BJ_B + j) * BJ_W + i];
          pv[t] = Qb[(size_t)Pa * tile + (ha * BJ_B + j) * BJ_W + i];
        }
      }
      S[x * BJ_ILD + y] = fI ? dIs : dIa;
      S[(BJ_B + x) * BJ_ILD + BJ_B + y] = fJ ? dJs : dJa;
      if (PI == PJ || (!fI && !fJ)) {
        const real v = PI == PJ ? (fI ? cs_ : ca_) : ca_;
        S[x * BJ_ILD + BJ_B + y] = v;
        S[(BJ_B + y) * BJ_ILD + x] = v;
      } else {
        // the tile (Pa <= Pb) as bj_update_job forms it: O = Q_Pa' A[Pa, Pb] Q_Pb, of which the rows of half ha and the columns of half hb
        const int fa = swap ? fJ : fI, fb = swap ? fI : fJ;
        real *xs = S + BJ_W * BJ_ILD; // scratch behind S (see below)
        real *qq = xs + BJ_W * BJ_LD, *qp = qq + BJ_B * BJ_LD, *ts = qp + BJ_B * BJ_LD, *out = ts + BJ_B * BJ_LD;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int e = tid + t * BJ_UPD_THREADS, i = e & (BJ_W - 1), kk = e >> 6;
          xs[i * BJ_LD + kk] = xv[t];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int e = tid + t * BJ_UPD_THREADS, i = e & (BJ_W - 1), j = e >> 6;
          qq[j * BJ_LD + i] = fb ? qv[t] : (i == hb * BJ_B + j ? (real)1 : (real)0);
          qp[j * BJ_LD + i] = fa ? pv[t] : (i == ha * BJ_B + j ? (real)1 : (real)0);
        }
        __syncthreads();
        bj_gemm_part(xs, qq, ts, 4, 2, tid, nthr);  // T[:, hb half] = X Q_Pb[:, hb half]
        __syncthreads();
        bj_gemm_part(qp, ts, out, 2, 2, tid, nthr); // O[ha half, hb half] = Q_Pa[:, ha half]' T
        __syncthreads();
        const real v = out[y * BJ_LD + x]; // O[i = x][j = y]: row x of half ha, column y of half hb
        const int xx = swap ? y : x, yy = swap ? x : y; // xx: index in I', yy: index in J'
        S[xx * BJ_ILD + BJ_B + yy] = v;
        S[(BJ_B + yy) * BJ_ILD + xx] = v;
      }
    } else {
    const int fI = flags[PI], fJ = flags[PJ];
    // entry (hr * 32 + x, hc * 32 + y) of the diagonal tile of step-k pair P after the update: S' of the pair, or A itself if it did not rotate
    auto diag = [&](int P, int2 pr, int f, int rr, int cc) -> real {
      return f ? Sb[(size_t)P * BJ_W * BJ_W + cc * BJ_W + rr] : Aold[(size_t)bj_gidx(pr, cc) * ld + bj_gidx(pr, rr)];
    };
    for (int e = tid; e < 2 * BJ_B * BJ_B; e += nthr) { // quadrants [I', I'] and [J', J']
      const int x = e & (BJ_B - 1), y = (e >> 5) & (BJ_B - 1), which = e >> 10;
      if (which == 0) S[x * BJ_ILD + y] = diag(PI, prI, fI, hI * BJ_B + x, hI * BJ_B + y);
      else S[(BJ_B + x) * BJ_ILD + BJ_B + y] = diag(PJ, prJ, fJ, hJ * BJ_B + x, hJ * BJ_B + y);
    }
    if (PI == PJ) { // I' and J' sat in ONE pair of step k: the quadrant is a piece of that pair's diagonal tile too
      for (int e = tid; e < BJ_B * BJ_B; e += nthr) {
        const int x = e & (BJ_B - 1), y = e >> 5;
        const real v = diag(PI, prI, fI, hI * BJ_B + x, hJ * BJ_B + y);
        S[x * BJ_ILD + BJ_B + y] = v;
        S[(BJ_B + y) * BJ_ILD + x] = v;
      }
    } else if (!fI && !fJ) { // neither pair rotated: the update copies the tile
      for (int e = tid; e < BJ_B * BJ_B; e += nthr) {
        const int x = e & (BJ_B - 1), y = e >> 5;
        const real v = Aold[(size_t)bj_gidx(IJ, BJ_B + y) * ld + bj_gidx(IJ, x)];
        S[x * BJ_ILD + BJ_B + y] = v;
        S[(BJ_B + y) * BJ_ILD + x] = v;
      }
    } else {
      // the tile (Pa <= Pb) as bj_update_job forms it: O = Q_Pa' A[Pa, Pb] Q_Pb, of which the rows of half ha and the columns of half hb
      const bool swap = PI > PJ;
      const int Pa = swap ? PJ : PI, Pb = swap ? PI : PJ, ha = swap ? hJ : hI, hb = swap ? hI : hJ, fa = swap ? fJ : fI, fb = swap ? fI : fJ;
      const int2 pra = swap ? prJ : prI, prb = swap ? prI : prJ;
      real *xs = S + BJ_W * BJ_ILD; // scratch behind S, over its second copy, Q and the rotation tables (which the sweep sets up afterwards): ends at 134 656 B
      static_assert((size_t)(BJ_W * BJ_ILD + BJ_W * BJ_LD + 4 * BJ_B * BJ_LD) * sizeof(real) <= BJ_UPDATE_LDS, "k_bj_fused: prologue scratch");
      real *qq = xs + BJ_W * BJ_LD, *qp = qq + BJ_B * BJ_LD, *ts = qp + BJ_B * BJ_LD, *out = ts + BJ_B * BJ_LD;
      for (int e = tid; e < BJ_W * BJ_W; e += nthr) {
        const int i = e & (BJ_W - 1), kk = e >> 6;
        xs[i * BJ_LD + kk] = Aold[(size_t)bj_gidx(prb, kk) * ld + bj_gidx(pra, i)];
      }
      for (int e = tid; e < BJ_W * BJ_B; e += nthr) { // 32 columns of each Q (identity where the pair did not rotate)
        const int i = e & (BJ_W - 1), j = e >> 6;
        qq[j * BJ_LD + i] = fb ? Qb[(size_t)Pb * BJ_W * BJ_W + (hb * BJ_B + j) * BJ_W + i] : (i == hb * BJ_B + j ? (real)1 : (real)0);
        qp[j * BJ_LD + i] = fa ? Qb[(size_t)Pa * BJ_W * BJ_W + (ha * BJ_B + j) * BJ_W + i] : (i == ha * BJ_B + j ? (real)1 : (real)0);
      }
      __syncthreads();
      bj_gemm_part(xs, qq, ts, 4, 2, tid, nthr);  // T[:, hb half] = X Q_Pb[:, hb half]
      __syncthreads();
      bj_gemm_part(qp, ts, out, 2, 2, tid, nthr); // O[ha half, hb half] = Q_Pa[:, ha half]' T
      __syncthreads();
      for (int e = tid; e < BJ_B * BJ_B; e += nthr) {
        const int i = e & (BJ_B - 1), j = e >> 5; // O[i][j] at out[j * BJ_LD + i]: row i of half ha, column j of half hb
        const real v = out[j * BJ_LD + i];
        const int x = swap ? j : i, y = swap ? i : j; // x: index in I', y: index in J'
        S[x * BJ_ILD + BJ_B + y] = v;
        S[(BJ_B + y) * BJ_ILD + x] = v;
      }
    }
    }
  }
  __syncthreads();
  const size_t slot_q = (size_t)b * npmax + pi;
  bj_inner_sweep(bj_smem, IJ, kind, sh.k, thr, ctl, (ctl->sweeps + (F.next_sweep ? 1 : 0)) & 1, Qbuf2 + F.qout * qsz + slot_q * BJ_W * BJ_W,
                 Sbuf2 + F.qout * qsz + slot_q * BJ_W * BJ_W, Qflag2 + F.qout * fsz + slot_q);
}

// closes a sweep for every block (one wave, a lane per block); remaining[0] = blocks still iterating.  scanned: left_bits holds the largest
// off-diagonal entry of the matrix as this sweep left it
__global__ __launch_bounds__(SCSAMD_WAVE) void k_bp_sweep_end(BigPsdView B, int *status, int *remaining, int scanned) {
  int rem = 0, worked = 0;
  for (int b = threadIdx.x; b < B.nbig; b += SCSAMD_WAVE) {
    BigPsdCtl *c = B.ctl + b;
    if (c->done) continue;
    worked = 1;
    const int par = c->sweeps & 1;
    c->sweeps += 1;
    const bool nothing_left = scanned && bp_from_bits(c->left_bits) <= c->thr; // the next sweep would rotate nothing
    c->left_bits = 0ull;
    if (bp_from_bits(c->offmax_bits[par]) <= c->thr || nothing_left) {
      c->done = 1;
    } else if (c->sweeps >= PSD_MAX_SWEEPS) {
      c->done = 1;
      atomicAdd(status, 1); // did not converge: counted, not fatal (cones.c:1031-1032)
    } else {
      ++rem;
    }
    c->offmax_bits[par] = 0ull;
  }
  rem = wave_sum(rem);
  worked = wave_max(worked);
  if (threadIdx.x == 0) {
    remaining[0] = rem;
    remaining[1] += worked; // sweeps of this projection in which some block still iterated (the host sizes its next batch from it)
  }
}

// The largest off-diagonal entry of every block that is still iterating, in the matrix as the sweep just run leaves it (round 6).
// The iteration ends with a sweep that meets no entry above the threshold, i.e. one that rotates nothing: whether the NEXT sweep
// would be that one can be read off the matrix in one pass (8 MB at order 1024) instead of being found out by running it (32
// launches of ~20 us there, 8 at order 256: 5 % of a cold projection, a third to a half of a warm-started one of 2 - 3 sweeps).
// Same rule (|a_pq| > thr rotates, both indices below k), same final matrix bit for bit; both triangles are read, so whichever mirror
// entry a rotation would have looked at is covered.  slot: parity of the next launch (ctl.cur), as k_bp_scale.
// The fused step does this inside the last update of the sweep (bj_update_job's left_bits: that launch rewrites the whole matrix anyway);
// this kernel serves the forms that do not come through there (two-launch step, single-column steps, option psd_offscan = 2).
// (Closing the sweep from the workgroup that finishes last -- a ticket behind a fence -- instead of k_bp_sweep_end's launch was
// measured and dropped: an agent-scope fence per workgroup writes the XCD's L2 back; 100 x 32: 1.73 vs 1.35 ms per projection.)
__global__ __launch_bounds__(BP_THREADS) void k_bp_offscan(BigPsdView B, int slot) {
  __shared__ real red[BP_THREADS / SCSAMD_WAVE];
  const int b = blockIdx.y;
  BigPsdCtl *c = B.ctl + b;
  if (c->done) return;
  const BlockShape s = bp_shape_raw(c->kraw);
  const real *A = (c->cur[slot] ? B.A1 : B.A) + (size_t)b * B.ld * B.ld;
  real mx = 0;
  for (int j = blockIdx.x; j < s.k; j += gridDim.x) {
    const real *col = A + (size_t)j * B.ld;
    for (int i = threadIdx.x; i < s.k; i += BP_THREADS) {
      const real a = absval(col[i]);
      if (i != j && a > mx) mx = a;
    }
  }
  mx = block_max(mx, red);
  // (the relaxed read may be behind the other workgroups' maxima, i.e. too small: a needless atomic then, never a missing one)
  if (threadIdx.x == 0 && mx > bp_from_bits(__hip_atomic_load(&c->left_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicMax(&c->left_bits, bp_bits(mx));
}

// W = V diag(sqrt(max(lambda, 0)))   (cones.c:1036-1044), in place
__global__ __launch_bounds__(BP_THREADS) void k_bp_scale(BigPsdView B, int slot) {
  const int b = blockIdx.y;
  const BlockShape s = bp_shape(B, b);
  const real *A = (B.ctl[b].cur[slot] ? B.A1 : B.A) + (size_t)b * B.ld * B.ld; // the copy the last step left the block in
  real *V = B.V + (size_t)b * B.ld * B.ld;
  const long long total = (long long)s.K2 * s.K2;
  for (long long e = (long long)blockIdx.x * BP_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * BP_THREADS) {
    const int i = (int)(e % s.K2), c = (int)(e / s.K2);
    const real lam = A[(size_t)c * B.ld + c];
    V[(size_t)c * B.ld + i] *= (c < s.k && lam > (real)0) ? sqrt(lam) : (real)0;
  }
}

// X+ = W W', packed lower triangle with diagonal / sqrt(2) (cones.c:1052-1063): real blocks on the fp64 matrix cores,
// 16x16 MFMA tiles (lane l supplies W[row l&15][k l>>4] for both operands; D element (row (l>>4) + 4 reg, col l&15)).  Complex blocks and the fp32 build: one lane per packed output entry.
__global__ __launch_bounds__(BP_THREADS) void k_bp_gram(BigPsdView B, real *x) {
  const int b = blockIdx.y;
  const BlockShape s = bp_shape(B, b);
  real *X = x + B.psd_off[B.id[b]];
  const real *W = B.V + (size_t)b * B.ld * B.ld;
  const size_t ld = B.ld;
  const real inv_sqrt2 = (real)1 / sqrt((real)2);
  const int k = s.k;
  if (s.cplx) { // Hermitian repack (cones.c:1139-1145): Re = (W W')[r][c], Im = (W W')[r+nn][c]
    const int nn = s.nn;
    for (long long e = (long long)blockIdx.x * BP_THREADS + threadIdx.x; e < (long long)nn * nn; e += (long long)gridDim.x * BP_THREADS) {
      int c = 0;
      long long rem = e;
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
        const int r = c + 1 + (int)((rem - 1) / 2);
        ra = ((rem - 1) & 1) ? r + nn : r;
      }
      real acc = 0;
      for (int cc = 0; cc < k; ++cc) acc += W[cc * ld + ra] * W[cc * ld + rb];
      X[e] = acc * scale;
    }
    return;
  }
#ifndef SFLOAT
  // (round 6) a wave owns a 32 x 32 tile of the lower triangle: 2 x 2 MFMA tiles, every fragment of W feeds two products; the tile above
  // the diagonal of a diagonal 32 x 32 tile is not formed.  Additions into every entry in the order of rounds 2-5 (same bits).
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = BP_THREADS >> 6;
  const int T = (k + 31) >> 5;
  const long long ntile = (long long)T * (T + 1) / 2;
  const int ksteps = (s.K2 + 3) >> 2;
  const int li = lane & 15, lk = lane >> 4;
  for (long long t = (long long)blockIdx.x * nw + wave; t < ntile; t += (long long)gridDim.x * nw) {
    // t -> (ti, tj), tj <= ti, rows of tiles enumerated one after the other
    int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long long)ti * (ti + 1) / 2 > t) --ti;
    while ((long long)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = (int)(t - (long long)ti * (ti + 1) / 2);
    const bool diag_tile = ti == tj;
    f64x4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    const int ra = ti * 32 + li, rb = tj * 32 + li;
    // eight steps (32 summation indices, 28 - 32 products) per trip, the next trip's fragments asked for before this one's products
    // (one step ahead was measured: 179 us at order 1024 against 118 for round 5's 16 x 16 tiles -- the L2 latency showed whole)
    constexpr int GS = 8;
    auto frag = [&](int ks0, double (&av)[GS][2], double (&bv)[GS][2]) {
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        const int kc = (ks0 + u) * 4 + lk;
        const bool kin = kc < s.K2; // (beyond the last step: zeros, not used)
        const real *col = W + (size_t)kc * ld;
        av[u][0] = (kin && ra < k) ? col[ra] : 0.0;
        av[u][1] = (kin && ra + 16 < k) ? col[ra + 16] : 0.0;
        bv[u][0] = (kin && rb < k) ? col[rb] : 0.0;
        bv[u][1] = (kin && rb + 16 < k) ? col[rb + 16] : 0.0;
      }
    };
    double av[GS][2], bv[GS][2], an[GS][2], bn[GS][2];
    frag(0, av, bv);
    for (int ks = 0; ks < ksteps; ks += GS) {
      frag(ks + GS, an, bn);
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        if (ks + u < ksteps) { // (uniform)
          acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][0], bv[u][0], acc[0][0], 0, 0, 0);
          if (!diag_tile) acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][0], bv[u][1], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][1], bv[u][0], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][1], bv[u][1], acc[1][1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < GS; ++u) av[u][0] = an[u][0], av[u][1] = an[u][1], bv[u][0] = bn[u][0], bv[u][1] = bn[u][1];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ti * 32 + x * 16 + lk + 4 * r, j = tj * 32 + y * 16 + li;
          if (i < k && j <= i) X[packed_index(i, j, k)] = i == j ? acc[x][y][r] * inv_sqrt2 : acc[x][y][r];
        }
  }
#else
  const long long ntri = (long long)k * (k + 1) / 2;
  for (long long e = (long long)blockIdx.x * BP_THREADS + threadIdx.x; e < ntri; e += (long long)gridDim.x * BP_THREADS) {
    int j = 0;
    long long rem = e;
    while (rem >= k - j) {
      rem -= k - j;
      ++j;
    }
    const int i = j + (int)rem;
    real acc = 0;
    for (int cc = 0; cc < k; ++cc) acc += W[cc * ld + i] * W[cc * ld + j];
    if (i == j) acc *= inv_sqrt2;
    X[e] = acc;
  }
#endif
}

__global__ void k_bp_set_kraw(BigPsdView B, int *remaining) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B.nbig) B.ctl[b].kraw = B.psd_k[B.id[b]];
  if (b == 0) remaining[1] = 0;
}

// C = L' R for K2 x K2 column-major matrices (both operands contiguous along the summation index): the two products
// of the warm start A' = Vp' (A Vp) -- A is symmetric, so A Vp = A' Vp.  fp64 matrix cores; lane (li = l & 15, lk = l >> 4) owns
// the summation indices 16 kk + 4 lk + s for MFMA step s of chunk kk (any assignment works as long as both operands use the same
// one), i.e. one 32-byte run per operand and chunk, so a wave reads whole 128-byte lines.
// Round 6: a wave owns a 32 x 32 tile of C (2 x 2 MFMA tiles: every operand fragment feeds two products, half the L2 traffic per
// flop) and asks for chunk kk + 1 before it multiplies chunk kk.  Rounds 2-5: one 16 x 16 tile per wave, loads and products in turn --
// 234 us per product at order 1024 (9 TFLOP/s).  The order of the additions into every entry of C is unchanged (same bits).
// fp32 build: plain loops.
__device__ __forceinline__ void bp_frag4(const real *__restrict__ M, size_t ld, int row, int k0, int K2, double (&f)[4]) {
#ifndef SFLOAT
  if (row < K2 && k0 + 3 < K2) { // (k0 is a multiple of 4 and ld is even: 16-byte aligned)
    const double2 lo = *reinterpret_cast<const double2 *>(M + (size_t)row * ld + k0), hi = *reinterpret_cast<const double2 *>(M + (size_t)row * ld + k0 + 2);
    f[0] = lo.x, f[1] = lo.y, f[2] = hi.x, f[3] = hi.y;
    return;
  }
#endif
#pragma unroll
  for (int q = 0; q < 4; ++q) f[q] = (row < K2 && k0 + q < K2) ? (double)M[(size_t)row * ld + k0 + q] : 0.0;
}
__global__ __launch_bounds__(BP_THREADS) void k_bp_gemm_tn(BigPsdView B, real *C, const real *__restrict__ L, const real *__restrict__ R) {
  const int b = blockIdx.y;
  const BlockShape s = bp_shape_raw(B.ctl[b].kraw);
  const size_t ld = B.ld, mat = (size_t)b * ld * ld;
  const int K2 = s.K2;
  C += mat;
  L += mat;
  R += mat;
#ifndef SFLOAT
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = BP_THREADS >> 6;
  const int T = (K2 + 31) >> 5, li = lane & 15, lk = lane >> 4;
  const int chunks = (K2 + 15) >> 4;
  for (long long t = (long long)blockIdx.x * nw + wave; t < (long long)T * T; t += (long long)gridDim.x * nw) {
    const int ti = (int)(t % T), tj = (int)(t / T);
    const int ra = ti * 32 + li, cb = tj * 32 + li;
    f64x4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    // two chunks (32 summation indices, 32 products = ~2000 clocks of the matrix core) per trip, the next trip's fragments asked for
    // before the products of this one: with one wave per SIMD (order 1024: 1024 tiles) nothing else hides the L2 latency
    double a[2][2][4], bb[2][2][4], an[2][2][4], bn[2][2][4]; // [chunk of the trip][tile row / column][step]
    auto trip = [&](int kk, double (&fa)[2][2][4], double (&fb)[2][2][4]) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int k0 = (kk + c) * 16 + lk * 4; // (beyond the last chunk: zeros, not used)
        bp_frag4(L, ld, ra, k0, K2, fa[c][0]);
        bp_frag4(L, ld, ra + 16, k0, K2, fa[c][1]);
        bp_frag4(R, ld, cb, k0, K2, fb[c][0]);
        bp_frag4(R, ld, cb + 16, k0, K2, fb[c][1]);
      }
    };
    trip(0, a, bb);
    for (int kk = 0; kk < chunks; kk += 2) {
      trip(kk + 2, an, bn);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (kk + c < chunks) { // (uniform)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
              for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][x][q], bb[c][y][q], acc[x][y], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int q = 0; q < 4; ++q) a[c][x][q] = an[c][x][q], bb[c][x][q] = bn[c][x][q];
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ti * 32 + x * 16 + lk + 4 * r, j = tj * 32 + y * 16 + li;
          if (i < K2 && j < K2) C[(size_t)j * ld + i] = acc[x][y][r];
        }
  }
#else
  for (long long e = (long long)blockIdx.x * BP_THREADS + threadIdx.x; e < (long long)K2 * K2; e += (long long)gridDim.x * BP_THREADS) {
    const int i = (int)(e % K2), j = (int)(e / K2);
    real acc = 0;
    for (int kc = 0; kc < K2; ++kc) acc += L[i * ld + kc] * R[j * ld + kc];
    C[j * ld + i] = acc;
  }
#endif
}

// A <- (A + A') / 2 (the rotations assume exact symmetry)
__global__ __launch_bounds__(BP_THREADS) void k_bp_symm(BigPsdView B) {
  const int b = blockIdx.y;
  const BlockShape s = bp_shape_raw(B.ctl[b].kraw);
  real *A = B.A + (size_t)b * B.ld * B.ld;
  const size_t ld = B.ld;
  for (long long e = (long long)blockIdx.x * BP_THREADS + threadIdx.x; e < (long long)s.K2 * s.K2; e += (long long)gridDim.x * BP_THREADS) {
    const int i = (int)(e % s.K2), j = (int)(e / s.K2);
    if (i > j) {
      const real v = (real)0.5 * (A[j * ld + i] + A[i * ld + j]);
      A[j * ld + i] = v;
      A[i * ld + j] = v;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct BigPsd {
  int nbig = 0, kmax = 0, ld = 0;
  DevBuf<int> id;
  DevBuf<real> A, V, Vp, Tm; // working matrix (copy 0), eigenvectors / W, carried eigenbasis, warm-start temporary = copy 1 of A
  bool have_basis = false;
  long long calls = 0;
  DevBuf<BigPsdCtl> ctl;
  DevBuf<int> remaining;
  PinnedBuf<int> hrem;       // host copy of `remaining`
  long long sweeps_total = 0, projections = 0;
  bool warm_ok = true;
  bool blocked = true;       // tournament over 32-wide block columns with MFMA updates (k_bj_*); false: single columns (k_bp_step)
  int sweeps_hint[2] = {0, 0}; // sweeps the previous projection of the same kind ([0] cold start, [1] warm start) needed: that many minus one
                             // are enqueued before the first read-back (a cold projection's ~10 must not size the next warm one's batch)
  bool fused = true;         // blocked: one launch per outer step (k_bj_fused) instead of k_bj_inner + k_bj_update
  DevBuf<real> normpart;     // k_bp_norm_part's partial sums
  bool offscan_fold = true;  // the scan inside the last update of a sweep (fused step) instead of k_bp_offscan's launch
  bool flat_prologue = true; // k_bj_fused: the inner sweep's subproblem from one level of global loads
  bool grid_1d = true;       // k_bj_fused: one-dimensional grid, inner-sweep workgroups of all blocks first
  bool offscan = true;       // a pass over the matrix after every sweep decides whether another sweep would rotate anything (k_bp_offscan)
  bool cross = true;         // blocked: every index pair once per sweep (within pass + cross-pair tournament steps), see bj_pair_sched
  DevBuf<real> Qbuf, Sbuf;   // blocked: per (block, block-column pair) the 64 x 64 factor Q and rotated subproblem S'
  DevBuf<int> Qflag;         // ... and whether the pair's inner sweep rotated at all
  void reset_warm_start() { calls = 0; have_basis = false; sweeps_hint[0] = sweeps_hint[1] = 0; }

  // pk: signed orders of all PSD blocks (negative = complex embedding order); blocks above lds_kmax are taken
  void init(const std::vector<int> &pk, int lds_kmax, hipStream_t st) {
    std::vector<int> ids;
    kmax = 0;
    for (size_t i = 0; i < pk.size(); ++i) {
      const int ka = pk[i] < 0 ? -pk[i] : pk[i];
      if (ka > lds_kmax) {
        ids.push_back((int)i);
        kmax = std::max(kmax, ka);
      }
    }
    nbig = (int)ids.size();
    if (!nbig) return;
    blocked = true;
    if (const char *e = opt_get("psd_blocked")) blocked = atoi(e) != 0; // 0: the single-column steps of round 2 (A/B measurements)
    ld = blocked ? (kmax + BJ_W - 1) / BJ_W * BJ_W : (kmax + 1) & ~1;
    normpart.alloc((size_t)nbig * BP_NORM_G);
    offscan_fold = true;
    if (const char *e = opt_get("psd_offscan")) offscan_fold = atoi(e) != 2; // 2: the scan as a launch of its own (A/B measurements)
    flat_prologue = true;
    if (const char *e = opt_get("psd_prologue")) flat_prologue = atoi(e) != 0; // 0: rounds 4-5's three dependent levels of loads (A/B measurements)
    grid_1d = true;
    if (const char *e = opt_get("psd_grid")) grid_1d = atoi(e) != 0; // 0: (job, block) grid of rounds 4-5 (A/B measurements)
    offscan = true;
    if (const char *e = opt_get("psd_offscan")) offscan = atoi(e) != 0; // 0: rounds 2-5's closing sweep that rotates nothing (A/B measurements)
    cross = true;
    if (const char *e = opt_get("psd_cross")) cross = atoi(e) != 0; // 0: full 63-step sweeps of every block-column pair (first form of round 3)
    if (blocked) {
      const size_t npmax = (size_t)ld / BJ_W;
      fused = true;
      if (const char *e = opt_get("psd_fused")) fused = atoi(e) != 0; // 0: inner sweep and update as two launches per outer step (A/B measurements)
      Qbuf.alloc((size_t)2 * nbig * npmax * BJ_W * BJ_W); // two generations (k_bj_fused)
      Sbuf.alloc((size_t)2 * nbig * npmax * BJ_W * BJ_W);
      Qflag.alloc((size_t)2 * nbig * npmax);
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bj_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BJ_UPDATE_LDS));
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bj_inner), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BJ_INNER_LDS));
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bj_update), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BJ_UPDATE_LDS));
    }
    id.alloc(ids.size());
    id.upload(ids.data(), ids.size(), st);
    A.alloc((size_t)nbig * ld * ld);
    V.alloc((size_t)nbig * ld * ld);
    warm_ok = !opt_get("psd_cold");
    if (warm_ok) Vp.alloc((size_t)nbig * ld * ld);
    Tm.alloc((size_t)nbig * ld * ld); // T of the warm start, then the second copy of A during the sweeps
    have_basis = false;
    calls = 0;
    ctl.alloc(nbig);
    remaining.alloc(2);
    hrem.alloc(2);
    HIP_CHECK(hipStreamSynchronize(st));
  }

  // x: the cone vector (device); status: the ConeDev sweep-cap counter
  void project(real *x, const int *psd_off, const int *psd_k, int *status, hipStream_t st) {
    if (!nbig) return;
    BigPsdView B{nbig, ld, id.p, psd_off, psd_k, A.p, V.p, Tm.p, ctl.p};
    const long long elems = (long long)ld * ld;
    const int g_elem = (int)std::min<long long>((elems + BP_THREADS - 1) / BP_THREADS, 2048);
    const int np_max = ld / 2, TPm = (np_max + BP_TILE - 1) / BP_TILE;
    const int g_step = TPm * TPm + ((np_max + BP_VPAIRS - 1) / BP_VPAIRS) * ((ld + BP_VROWS - 1) / BP_VROWS); // tiles of the largest block
    // warm start (as in the LDS kernel): iterate on A' = Vp' A Vp from V = Vp, the eigenbasis of the previous projection
    // of the same block; a cold restart every PSD_WARM_RESET calls bounds the orthogonality drift of the carried basis
    const bool warm = warm_ok && have_basis && (calls % PSD_WARM_RESET) != 0;
    ++calls;
    const size_t mat_bytes = (size_t)nbig * ld * ld * sizeof(real);
    hipLaunchKernelGGL(k_bp_set_kraw, dim3((nbig + 63) / 64), dim3(64), 0, st, B, remaining.p);
    hipLaunchKernelGGL(k_bp_unpack, dim3(g_elem, nbig), dim3(BP_THREADS), 0, st, B, x, warm ? 0 : 1, blocked ? 1 : 0);
    if (warm) {
      const long long T32 = (ld + 31) / 32;
      const int g_mm = (int)std::min<long long>((T32 * T32 + 3) / 4, 8192); // one 32 x 32 tile per wave
      HIP_CHECK(hipMemcpyAsync(V.p, Vp.p, mat_bytes, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(k_bp_gemm_tn, dim3(g_mm, nbig), dim3(BP_THREADS), 0, st, B, Tm.p, (const real *)A.p, (const real *)V.p); // T = A' Vp = A Vp
      hipLaunchKernelGGL(k_bp_gemm_tn, dim3(g_mm, nbig), dim3(BP_THREADS), 0, st, B, A.p, (const real *)V.p, (const real *)Tm.p); // A = Vp' T
      hipLaunchKernelGGL(k_bp_symm, dim3(g_elem, nbig), dim3(BP_THREADS), 0, st, B);
    }
    hipLaunchKernelGGL(k_bp_norm_part, dim3(BP_NORM_G, nbig), dim3(BP_THREADS), 0, st, B, normpart.p);
    hipLaunchKernelGGL(k_bp_norm, dim3(nbig), dim3(SCSAMD_WAVE), 0, st, B, (const real *)normpart.p);
    int *h_rem = hrem.p; // (pinned: the read-back of two ints into pageable memory went through the runtime's staging copy)
    h_rem[0] = nbig;
    h_rem[1] = 0;
    long long qgen = 0;  // fused step: generations of Q / S' / flags written so far
    bool first_launch = true;
    long long gstep = 0; // launches so far: the copy of A a block is in alternates with the steps IT took (ctl.cur)
    const long long sweeps_before = sweeps_total;
    const int nbc_max = ld / BJ_B, npmax = ld / BJ_W;
    const int g_upd = npmax * (npmax + 1) / 2 + npmax * (ld / BJ_W); // A tiles (P <= Q: the mirror tile is written by the same workgroup) + V tiles of the largest block
    // Sweeps are enqueued in batches: as many as the previous projection of these blocks needed, minus one, before the first
    // read-back, then one at a time (consecutive ADMM iterates need nearly the same count; a block that has converged makes every
    // later launch return at once, as in the PCG loop; the sweep cap of cones.c:1031 is enforced on the device).  One host
    // round trip per sweep cost 50 - 100 us -- a third of a projection of 32 blocks of order 100.
    const bool scan_in_update = offscan && blocked && fused && offscan_fold; // the fused step's last update of a sweep does the scan
    const int g_scan = std::max(1, std::min(ld / 2, std::max(ld / 8, 1024 / nbig))); // whole columns per workgroup; ~1024 workgroups in all (order 1024 on 64: 21 us)
    int enq = 0, batch = std::max(1, std::min(sweeps_hint[warm ? 1 : 0] - 1, warm ? 8 : 12)); // (all of them: measured the same, 2.70 vs 2.71 ms at 256 x 8)
    while (h_rem[0] > 0 && enq < PSD_MAX_SWEEPS + 4) {
     for (int bsw = 0; bsw < batch; ++bsw, ++enq) {
      if (blocked) {
        const int osteps = cross ? nbc_max : nbc_max - 1; // cross schedule: the within pass, then the tournament steps
        if (fused) {
          const int g_fused = npmax + g_upd;
          for (int step = 0; step < osteps; ++step, ++gstep) {
            if (first_launch) { // inner sweep of step 0 of the first sweep; nothing to update yet
              const BjFusedArgs F0{(int)(gstep & 1), 0, 0, 0, 0, 0, (int)(qgen & 1), cross ? 1 : 0, scan_in_update ? 1 : 0, flat_prologue ? 1 : 0, grid_1d ? 1 : 0};
              hipLaunchKernelGGL(k_bj_fused, grid_1d ? dim3(npmax * nbig) : dim3(npmax, nbig), dim3(BJ_UPD_THREADS), BJ_UPDATE_LDS, st, B, Qbuf.p, Sbuf.p, Qflag.p, npmax, F0);
              ++qgen;
              first_launch = false;
            }
            const bool last = step + 1 == osteps;
            const BjFusedArgs F{(int)(gstep & 1), 1, step, last ? 0 : step + 1, last ? 1 : 0, (int)((qgen - 1) & 1), (int)(qgen & 1), cross ? 1 : 0, scan_in_update ? 1 : 0, flat_prologue ? 1 : 0, grid_1d ? 1 : 0};
            hipLaunchKernelGGL(k_bj_fused, grid_1d ? dim3(g_fused * nbig) : dim3(g_fused, nbig), dim3(BJ_UPD_THREADS), BJ_UPDATE_LDS, st, B, Qbuf.p, Sbuf.p, Qflag.p, npmax, F);
            ++qgen;
          }
        } else
        for (int step = 0; step < osteps; ++step, ++gstep) {
          const int arg = (int)(gstep & 1) | (step << 1);
          hipLaunchKernelGGL(k_bj_inner, dim3(npmax, nbig), dim3(BJ_INNER_LAUNCH), BJ_INNER_LDS, st, B, Qbuf.p, Sbuf.p, Qflag.p, npmax, arg,
                             cross ? 1 : 0);
          hipLaunchKernelGGL(k_bj_update, dim3(g_upd, nbig), dim3(BJ_UPD_THREADS), BJ_UPDATE_LDS, st, B, (const real *)Qbuf.p,
                             (const real *)Sbuf.p, (const int *)Qflag.p, npmax, arg, cross ? 1 : 0);
        }
      } else {
        for (int step = 0; step < ld - 1; ++step, ++gstep)
          hipLaunchKernelGGL(k_bp_step, dim3(g_step, nbig), dim3(BP_THREADS), 0, st, B, (int)(gstep & 1) | (step << 1));
      }
      if (offscan && !scan_in_update) hipLaunchKernelGGL(k_bp_offscan, dim3(g_scan, nbig), dim3(BP_THREADS), 0, st, B, (int)(gstep & 1));
      hipLaunchKernelGGL(k_bp_sweep_end, dim3(1), dim3(SCSAMD_WAVE), 0, st, B, status, remaining.p, offscan ? 1 : 0);
     }
      HIP_CHECK(hipMemcpyAsync(h_rem, remaining.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      batch = 1;
    }
    sweeps_total += h_rem[1];
    sweeps_hint[warm ? 1 : 0] = h_rem[1];
    ++projections;
    static const bool debug = opt_get("debug") != nullptr;
    if (debug) fprintf(stderr, "[scs_amd psd_big] projection %lld: %s start, sweeps so far %lld (this one %lld)\n", projections, warm ? "warm" : "cold", sweeps_total, sweeps_total - sweeps_before);
    if (warm_ok) {
      HIP_CHECK(hipMemcpyAsync(Vp.p, V.p, mat_bytes, hipMemcpyDeviceToDevice, st));
      have_basis = true;
    }
    hipLaunchKernelGGL(k_bp_scale, dim3(g_elem, nbig), dim3(BP_THREADS), 0, st, B, (int)(gstep & 1));
    const long long T = (kmax + 31) / 32, ntile = T * (T + 1) / 2; // 32 x 32 tiles of the lower triangle, one per wave
    const long long outs = (long long)kmax * (kmax + 1) / 2;
    const int g_gram = (int)std::min<long long>(std::max<long long>((ntile + 3) / 4, (outs + 64LL * BP_THREADS - 1) / (64LL * BP_THREADS)), 8192);
    hipLaunchKernelGGL(k_bp_gram, dim3(std::max(1, g_gram), nbig), dim3(BP_THREADS), 0, st, B, x);
  }
};

} // namespace scsamd
