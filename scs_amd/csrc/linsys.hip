// linsys.hip -- Jacobi-preconditioned CG on the reduced KKT system, device
// resident and device controlled, plus the B1 host-pointer C ABI.
//
// What it computes is exactly reference linsys/cpu/indirect/private.c:
//   scs_solve_lin_sys :284-324   rhs fold-in, pcg, back-substitution for y
//   pcg               :133-217   stop on ||r||_inf < tol, early-out, breakdown
//   mat_vec           :106-119   y = R_x x + P x + A' R_y^-1 A x
//   set_preconditioner:50-82     M = 1 / diag(R_x + P + A' R_y^-1 A)
// How it runs is MI355X-first: four kernels per CG iteration
//   K1 csr_stream<DIV>   tmp = R_y^-1 (A p)
//   K2 csr_stream<GP>    Gp = R_x p + P p + A' tmp,  partials of p'Gp
//   K3 cg_update         alpha; x += alpha p; r -= alpha Gp; z = M r; partials z'r, |r|_inf
//   K4 cg_direction      convergence test; beta; p = z + beta p
// with all scalars (alpha, beta, z'r, ||r||, tol, done) living in HBM.  The host
// enqueues iterations in batches sized from the previous solve's count and reads
// one control block per batch; kernels issued past convergence return at once,
// so the iterate returned is exactly the first one meeting the tolerance.
#include "linsys.h"
#include <thread>
#include <exception>
#include <system_error>
#include <algorithm>
#include "spmv_wave_build.h"
#include "host_transpose.h"
#include <algorithm>
#include <atomic>

namespace scsamd {

constexpr int CG_GRAPH_ITERS = 8;               // iterations per captured graph (even: slot parity q = iteration & 1)
constexpr long long CG_GRAPH_MAX_NNZ = 2000000; // systems up to this many nonzeros replay the CG loop from a graph


constexpr int VEC_MAX_GRID = 512; // every consumer workgroup re-reduces the producers' partials: 2048 WGs cost K4 64 MB of L2 reads (measured 2.5 % of a CG iteration)
constexpr int PART_CAP = 4096; // >= SPMV_MAX_GRID, >= WR_MAX_GRID and >= 2 * VEC_MAX_GRID
static_assert(PART_CAP >= SPMV_MAX_GRID && PART_CAP >= WR_MAX_GRID && PART_CAP >= 2 * 512, "partial arrays too small");

CsrDev::~CsrDev() { delete wave; }

static inline int vec_grid(long long len) {
  // read once per process (called per launch: no table lookups here); tests that shrink it run in their own process
  static const int cap = [] {
    const char *e = opt_get("vec_max_grid"); // tests shrink it to force grid-striding
    int g = e ? atoi(e) : VEC_MAX_GRID;
    return (g >= 1 && g <= PART_CAP / 2) ? g : VEC_MAX_GRID; // z'r and |r| partials share one PART_CAP array (measurement sweeps go to 2048)
  }();
  long long g = (len + SCSAMD_BLOCK - 1) / SCSAMD_BLOCK;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ----------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_absmax_partial(const real *__restrict__ v, int len,
                                                                 real *part) {
  __shared__ real red[4];
  real mx = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
    real a = absval(v[i]);
    mx = a > mx ? a : mx;
  }
  mx = block_max(mx, red);
  if (threadIdx.x == 0) part[blockIdx.x] = mx;
}

// private.c:296-303: zero-rhs short circuit, tmp = R_y^-1 r_y; also arms the
// control block for this solve (and, optionally, forms the tolerance of
// src/scs.c:745-762 from the warm-start norm partials).
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_rhs_prep(real *b, const real *__restrict__ ry,
                                                           real *tmp, int n, int m,
                                                           const real *part, int pcount, CgCtl *ctl,
                                                           real tol, const real *warm_part,
                                                           int warm_cnt, real warm_scale, int max_its) {
  __shared__ real red[4];
  const real nb = reduce_partials_max(part, pcount, red);
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gs = gridDim.x * blockDim.x;
  const bool zero = nb <= (real)1e-12;
  if (zero) {
    for (int i = gtid; i < n + m; i += gs) b[i] = 0;
  } else {
    for (int i = gtid; i < m; i += gs) tmp[i] = b[n + i] / ry[i];
  }
  real t = tol;
  if (warm_part) {
    const real nw = reduce_partials_max(warm_part, warm_cnt, red) * warm_scale;
    t = t < nw ? t : nw;
    t = (real)0.2 * t; // CG_TOL_FACTOR, include/glbopts.h:250
    t = t > (real)1e-12 ? t : (real)1e-12; // CG_BEST_TOL, glbopts.h:247
  }
  if (gtid == 0) {
    ctl->zero_rhs = zero ? 1 : 0;
    ctl->cg_done = zero ? 1 : 0;
    ctl->iters = 0;
    ctl->max_its = max_its;
    ctl->tol = t;
    ctl->rhs_norm = nb;
    ctl->norm_r = 0;
    ctl->ztr[0] = 0;
    ctl->ztr[1] = 0;
  }
}

// private.c:145-172 (residual, preconditioned residual, z'r, ||r||)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_cg_init(real *x, const real *__restrict__ s, real *r,
                                                          real *z, const real *__restrict__ M, int n,
                                                          real *part_ztr, real *part_max,
                                                          const CgCtl *ctl) {
  __shared__ real red[4];
  if (ctl->zero_rhs) return;
  real ztr = 0, mx = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    real ri;
    if (s) {
      ri = x[i] - r[i]; // r held G s; r = b - G s
      x[i] = s[i];
    } else {
      ri = x[i];
      x[i] = 0;
    }
    r[i] = ri;
    const real zi = ri * M[i];
    z[i] = zi;
    ztr += zi * ri;
    const real a = absval(ri);
    mx = a > mx ? a : mx;
  }
  ztr = block_sum(ztr, red);
  mx = block_max(mx, red);
  if (threadIdx.x == 0) {
    part_ztr[blockIdx.x] = ztr;
    part_max[blockIdx.x] = mx;
  }
}

// private.c:163 early-out with max(tol, 1e-12); p = z
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_cg_start(real *p, const real *__restrict__ z, int n,
                                                           const real *part_ztr, const real *part_max,
                                                           int pcount, CgCtl *ctl) {
  __shared__ real red[4];
  if (ctl->zero_rhs) return;
  const real ztr = reduce_partials_sum(part_ztr, pcount, red);
  const real nr = reduce_partials_max(part_max, pcount, red);
  const real tol = ctl->tol;
  const real thr = tol > (real)1e-12 ? tol : (real)1e-12;
  const bool conv = nr < thr;
  if (!conv)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = z[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->ztr[0] = ztr;
    ctl->norm_r = nr;
    if (conv) ctl->cg_done = 1;
  }
}

// private.c:181-197
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_cg_update(real *x, real *r, real *z,
                                                            const real *__restrict__ p,
                                                            const real *__restrict__ Gp,
                                                            const real *__restrict__ M, int n,
                                                            const real *part_pgp, int cnt_pgp,
                                                            real *part_ztr, real *part_max,
                                                            const CgCtl *ctl, int parity, int ntm) {
  __shared__ real red[4];
  // small systems are bound by chains of dependent reads (~1 us each: the operands were written by the previous
  // kernel on other CUs): the lane's first vector chunk, the control words and the partials are all requested
  // before anything is waited for -- one round trip instead of four
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gs = gridDim.x * blockDim.x;
  const int nv = n / RVW;
  const bool has = gtid < nv;
  rvec P0 = {}, G0 = {}, M0 = {}, X0 = {}, R0 = {};
  if (has) {
    P0 = ldv(p, gtid);
    G0 = ldv(Gp, gtid);
    M0 = ldv(M, gtid);
    X0 = ldv(x, gtid);
    R0 = ldv(r, gtid);
  }
  const int done = ctl->cg_done;
  const real ztr_in = ctl->ztr[parity];
  real psum = 0;
  for (int i = threadIdx.x; i < cnt_pgp; i += blockDim.x) psum += part_pgp[i];
  if (done) return;
  const real pgp = block_sum(psum, red); // = reduce_partials_sum(part_pgp, cnt_pgp, red)
  const real alpha = ztr_in / pgp;
  real ztr = 0, mx = 0;
  if (has) {
    rvec Z;
#pragma unroll
    for (int e = 0; e < RVW; ++e) {
      X0.v[e] += alpha * P0.v[e];
      const real ri = R0.v[e] + (-alpha) * G0.v[e];
      R0.v[e] = ri;
      const real zi = ri * M0.v[e];
      Z.v[e] = zi;
      ztr += zi * ri;
      const real a = absval(ri);
      mx = a > mx ? a : mx;
    }
    stv(x, gtid, X0);
    stv(r, gtid, R0);
    stv(z, gtid, Z);
  }
  int iv0 = gtid + gs;
  if (ntm & 4) { // two 16-byte chunks per lane and array in flight (measurement variant; same per-lane summation order)
    for (; iv0 + gs < nv; iv0 += 2 * gs) {
      const int ia = iv0, ib = iv0 + gs;
      const rvec Pa = ldv_nt(p, ia), Pb = ldv_nt(p, ib), Ga = ldv_nt(Gp, ia), Gb = ldv_nt(Gp, ib), Ma = ldv_nt(M, ia), Mb = ldv_nt(M, ib);
      rvec Xa = ldv_nt(x, ia), Xb = ldv_nt(x, ib), Ra = ldv_nt(r, ia), Rb = ldv_nt(r, ib), Za, Zb;
#pragma unroll
      for (int e = 0; e < RVW; ++e) {
        Xa.v[e] += alpha * Pa.v[e];
        const real ri = Ra.v[e] + (-alpha) * Ga.v[e];
        Ra.v[e] = ri;
        const real zi = ri * Ma.v[e];
        Za.v[e] = zi;
        ztr += zi * ri;
        const real a = absval(ri);
        mx = a > mx ? a : mx;
      }
#pragma unroll
      for (int e = 0; e < RVW; ++e) {
        Xb.v[e] += alpha * Pb.v[e];
        const real ri = Rb.v[e] + (-alpha) * Gb.v[e];
        Rb.v[e] = ri;
        const real zi = ri * Mb.v[e];
        Zb.v[e] = zi;
        ztr += zi * ri;
        const real a = absval(ri);
        mx = a > mx ? a : mx;
      }
      stv_nt(x, ia, Xa);
      stv_nt(r, ia, Ra);
      stv(z, ia, Za);
      stv_nt(x, ib, Xb);
      stv_nt(r, ib, Rb);
      stv(z, ib, Zb);
    }
  }
  for (int iv = iv0; iv < nv; iv += gs) { // 16 B per lane per array
    const rvec P = (ntm & 2) ? ldv_nt(p, iv) : ldv(p, iv), G = (ntm & 2) ? ldv_nt(Gp, iv) : ldv(Gp, iv), Mv = ntm ? ldv_nt(M, iv) : ldv(M, iv);
    rvec X = ntm ? ldv_nt(x, iv) : ldv(x, iv), R = ntm ? ldv_nt(r, iv) : ldv(r, iv), Z;
#pragma unroll
    for (int e = 0; e < RVW; ++e) {
      X.v[e] += alpha * P.v[e];
      const real ri = R.v[e] + (-alpha) * G.v[e];
      R.v[e] = ri;
      const real zi = ri * Mv.v[e];
      Z.v[e] = zi;
      ztr += zi * ri;
      const real a = absval(ri);
      mx = a > mx ? a : mx;
    }
    if (ntm) {
      stv_nt(x, iv, X);
      stv_nt(r, iv, R);
    } else {
      stv(x, iv, X);
      stv(r, iv, R);
    }
    if (ntm & 2) stv_nt(z, iv, Z);
    else stv(z, iv, Z);
  }
  for (int i = nv * RVW + gtid; i < n; i += gs) {
    const real pi = p[i], gi = Gp[i];
    x[i] += alpha * pi;
    const real ri = r[i] + (-alpha) * gi;
    r[i] = ri;
    const real zi = ri * M[i];
    z[i] = zi;
    ztr += zi * ri;
    const real a = absval(ri);
    mx = a > mx ? a : mx;
  }
  ztr = block_sum(ztr, red);
  mx = block_max(mx, red);
  if (threadIdx.x == 0) {
    part_ztr[blockIdx.x] = ztr;
    part_max[blockIdx.x] = mx;
  }
}

// private.c:202-214
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_cg_direction(real *p, const real *__restrict__ z, int n,
                                                               const real *part_ztr,
                                                               const real *part_max, int pcount,
                                                               CgCtl *ctl, int parity, int dmode) {
  __shared__ real red[4];
  // as in k_cg_update: everything this kernel reads is requested up front (one round trip, not four)
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gs = gridDim.x * blockDim.x;
  const int nv = n / RVW;
  const bool has = gtid < nv;
  rvec Z0 = {}, P0 = {};
  if (has) {
    Z0 = ldv(z, gtid);
    P0 = ldv(p, gtid);
  }
  const int done = ctl->cg_done;
  const real ztr_prev = ctl->ztr[parity], tol = ctl->tol;
  real zs = 0, ms = 0;
  for (int i = threadIdx.x; i < pcount; i += blockDim.x) {
    zs += part_ztr[i];
    const real v = part_max[i];
    ms = v > ms ? v : ms;
  }
  if (done) return;
  const real ztr = block_sum(zs, red); // = reduce_partials_sum(part_ztr, pcount, red)
  const real nr = block_max(ms, red);  // = reduce_partials_max(part_max, pcount, red)
  const bool conv = nr < tol;
  const bool brk = !conv && ztr_prev == (real)0;
  if (!conv && !brk) {
    const real beta = ztr / ztr_prev;
    if (has) {
#pragma unroll
      for (int e = 0; e < RVW; ++e) P0.v[e] = Z0.v[e] + beta * P0.v[e];
      stv(p, gtid, P0);
    }
    int iv0 = gtid + gs;
    if (dmode & 4) { // two chunks per lane in flight (measurement variant)
      for (; iv0 + gs < nv; iv0 += 2 * gs) {
        const rvec Za = (dmode & 1) ? ldv_nt(z, iv0) : ldv(z, iv0), Zb = (dmode & 1) ? ldv_nt(z, iv0 + gs) : ldv(z, iv0 + gs);
        rvec Pa = ldv(p, iv0), Pb = ldv(p, iv0 + gs);
#pragma unroll
        for (int e = 0; e < RVW; ++e) {
          Pa.v[e] = Za.v[e] + beta * Pa.v[e];
          Pb.v[e] = Zb.v[e] + beta * Pb.v[e];
        }
        stv(p, iv0, Pa);
        stv(p, iv0 + gs, Pb);
      }
    }
    for (int iv = iv0; iv < nv; iv += gs) {
      const rvec Z = (dmode & 1) ? ldv_nt(z, iv) : ldv(z, iv); // z is dead after this read: non-temporal keeps it out of the way (dmode bit 0)
      rvec P = ldv(p, iv);
#pragma unroll
      for (int e = 0; e < RVW; ++e) P.v[e] = Z.v[e] + beta * P.v[e];
      if (dmode & 2) stv_nt(p, iv, P);
      else stv(p, iv, P);
    }
    for (int i = nv * RVW + gtid; i < n; i += gs) p[i] = z[i] + beta * p[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->ztr[parity ^ 1] = ztr;
    ctl->norm_r = nr;
    if (!brk) ctl->iters += 1; // converged at i -> i+1 ; breakdown returns i (private.c:203,216)
    if (conv || brk || ctl->iters >= ctl->max_its) ctl->cg_done = 1;
  }
}

// ----------------------------------------------------------------------------
// Three kernels per CG iteration (round 5): K1, K2 as before, then ONE vector kernel that does what k_cg_update AND
// k_cg_direction did.  private.c:181-214 needs two reductions per iteration -- p'Gp for alpha, then z'r of the NEW residual for beta --
// and the second one is what forced a second launch.  It follows from quantities the K2 epilogue can sum while it has Gp in hand:
//     z_new' r_new = sum M (r - alpha Gp)^2 = z'r - 2 alpha (z'Gp) + alpha^2 (Gp' M Gp)
// (EPI_GP3 leaves the partials of p'Gp, z'Gp = sum M r Gp and Gp'MGp), so this kernel knows alpha AND beta before it touches a vector and
// streams  x += alpha p;  r -= alpha Gp;  z = M r;  p = z + beta p  in one pass -- z is never stored (8 instead of 11 vector passes).
// The expansion is used for beta ONLY: the z'r that the NEXT alpha divides is summed directly from the new residual in this pass (partials,
// as before), so its cancellation error (~eps z'r_old / z'r_new, relative) never accumulates; it perturbs the search direction at rounding
// level, like a different summation order.  The stop test of private.c:202 for the residual this launch produces is evaluated by the NEXT
// launch (from this one's |r| partials): the iterate returned is exactly the first one with |r|_inf < tol, the iteration count is the
// reference's; what it costs is one matrix product enqueued past convergence per solve (~1 % at ~118 iterations per solve).
// `it` = iterations completed before this launch.  Partials double-buffered by the parity of `it`.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_cg3_update(real *x, real *r, real *p, const real *__restrict__ Gp,
                                                             const real *__restrict__ M, int n, const real *part_pgp,
                                                             const real *part_d1, const real *part_d2, int cnt_sp,
                                                             const real *part_ztr_prev, const real *part_max_prev, int cnt_v,
                                                             real *part_ztr, real *part_max, CgCtl *ctl, int it) {
  __shared__ real red[4];
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gs = gridDim.x * blockDim.x;
  const int nv = n / RVW;
  const bool has = gtid < nv;
  rvec P0 = {}, G0 = {}, M0 = {}, X0 = {}, R0 = {};
  if (has) { // the lane's first chunk travels with the scalars (one round trip, as in k_cg_update)
    P0 = ldv(p, gtid);
    G0 = ldv(Gp, gtid);
    M0 = ldv_nt(M, gtid);
    X0 = ldv_nt(x, gtid);
    R0 = ldv_nt(r, gtid);
  }
  const int done = ctl->cg_done, max_its = ctl->max_its;
  const real tol = ctl->tol;
  real ztr = ctl->ztr[0];
  real s0 = 0, s1 = 0, s2 = 0, zs = 0, ms = 0;
  for (int i = threadIdx.x; i < cnt_sp; i += blockDim.x) {
    s0 += part_pgp[i];
    s1 += part_d1[i];
    s2 += part_d2[i];
  }
  if (it > 0)
    for (int i = threadIdx.x; i < cnt_v; i += blockDim.x) {
      zs += part_ztr_prev[i];
      const real v = part_max_prev[i];
      ms = v > ms ? v : ms;
    }
  if (done) return;
  const real pgp = block_sum(s0, red), d1 = block_sum(s1, red), d2 = block_sum(s2, red);
  bool conv = false;
  real nr = 0;
  if (it > 0) { // the stop test for the residual the previous launch produced (private.c:202)
    ztr = block_sum(zs, red);
    nr = block_max(ms, red);
    conv = nr < tol;
  }
  const bool brk = !conv && ztr == (real)0; // private.c:206
  const bool stop = conv || brk || it >= max_its;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (it > 0) {
      ctl->norm_r = nr;
      ctl->ztr[0] = ztr;
      ctl->iters = it; // converged after `it` iterations -> it; breakdown in iteration `it` -> it (private.c:203,216)
    }
    if (stop) ctl->cg_done = 1;
  }
  if (stop) return;
  const real alpha = ztr / pgp;
  const real ztr_new = ztr + alpha * (alpha * d2 - (real)2 * d1);
  const real beta = ztr_new / ztr;
  real zo = 0, mx = 0;
  if (has) {
#pragma unroll
    for (int e = 0; e < RVW; ++e) {
      X0.v[e] += alpha * P0.v[e];
      const real ri = R0.v[e] + (-alpha) * G0.v[e];
      R0.v[e] = ri;
      const real zi = ri * M0.v[e];
      P0.v[e] = zi + beta * P0.v[e];
      zo += zi * ri;
      const real a = absval(ri);
      mx = a > mx ? a : mx;
    }
    stv_nt(x, gtid, X0);
    stv_nt(r, gtid, R0);
    stv(p, gtid, P0);
  }
  for (int iv = gtid + gs; iv < nv; iv += gs) { // 16 B per lane per array; x, r, M are streamed once per iteration: non-temporal
    rvec P = ldv(p, iv);
    const rvec G = ldv(Gp, iv), Mv = ldv_nt(M, iv);
    rvec X = ldv_nt(x, iv), R = ldv_nt(r, iv);
#pragma unroll
    for (int e = 0; e < RVW; ++e) {
      X.v[e] += alpha * P.v[e];
      const real ri = R.v[e] + (-alpha) * G.v[e];
      R.v[e] = ri;
      const real zi = ri * Mv.v[e];
      P.v[e] = zi + beta * P.v[e];
      zo += zi * ri;
      const real a = absval(ri);
      mx = a > mx ? a : mx;
    }
    stv_nt(x, iv, X);
    stv_nt(r, iv, R);
    stv(p, iv, P);
  }
  for (int i = nv * RVW + gtid; i < n; i += gs) {
    const real pi = p[i];
    x[i] += alpha * pi;
    const real ri = r[i] + (-alpha) * Gp[i];
    r[i] = ri;
    const real zi = ri * M[i];
    p[i] = zi + beta * pi;
    zo += zi * ri;
    const real a = absval(ri);
    mx = a > mx ? a : mx;
  }
  zo = block_sum(zo, red);
  mx = block_max(mx, red);
  if (threadIdx.x == 0) {
    part_ztr[blockIdx.x] = zo;
    part_max[blockIdx.x] = mx;
  }
}

// test hook of the DLONG build (SCS_AMD_TEST_OFFSET_BIAS): entry positions += bias (see CsrDev::bias)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_add_bias(eoff *a, long long len, eoff bias) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) a[i] += bias;
}

// p'Gp partials (sharded solve: the dot product can only be taken after the all-reduce that completes Gp)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_dot_partial(const real *__restrict__ a, const real *__restrict__ b, int n,
                                                              real *part, const int *skip) {
  if (skip && *skip) return;
  __shared__ real red[4];
  real acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += a[i] * b[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_reciprocal(real *v, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = (real)1 / v[i];
}
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_zero_unless_rank0(real *v, int n, int rank, const int *skip) {
  if (rank == 0 || (skip && *skip)) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = 0;
}

// private.c:50-82, one lane per column of A (= row of CSR(A'))
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_precond(CsrView At, const real *__restrict__ rx,
                                                          const real *__restrict__ ry,
                                                          const real *__restrict__ pdiag, real *M) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < At.rows; i += gridDim.x * blockDim.x) {
    real acc = rx[i];
    for (eoff k = At.ptr[i]; k < At.ptr[i + 1]; ++k) {
      const real a = At.val[k];
      acc += a * a / ry[At.idx[k]];
    }
    if (pdiag) acc += pdiag[i];
    M[i] = (real)1 / acc;
  }
}


// ----------------------------------------------------------------------------
// Small systems (n <= CG2_N_MAX): TWO launches per CG iteration instead of four.  Below a few thousand unknowns a CG
// kernel costs ~3.5 us whatever it computes (it hands data to the next kernel), so the count of launches is the time.
// k_cg2_a does, in EVERY workgroup and for ALL n entries, the update and the direction step of the previous
// iteration (private.c:181-214; n is small, the redundant work is a few KB per workgroup, and no cross-workgroup
// reduction is needed because every workgroup owns the complete sums), keeps the new p in LDS, and runs its share of
// z = R_y^-1 A p gathering p from LDS; workgroup 0 alone stores x, r, p and the control words.  The transposed product
// with the p'Gp partials stays csr_stream_kernel<EPI_GP>.  The arithmetic -- including the shape of every partial
// sum: the virtual 256-lane workgroups of k_cg_update, their wave trees, the fixed-order re-reduction -- is exactly
// that of the four-kernel path, so the iterates are bit-identical (tests/test_linsys_gpu.py).
// p and r are double-buffered by iteration parity (every workgroup reads the old ones while workgroup 0 writes the new
// ones); x is touched by workgroup 0 only.
// ----------------------------------------------------------------------------
constexpr int CG2_N_MAX = 1024;              // every workgroup redoes the whole update: pays off while that is ONE batch of
                                             // loads (measured us per CG iteration, two-launch vs four-kernel: n=500 15.8,
                                             // n=1000 16.8 vs 18.9, n=2000 19.2 vs 19.0, n=3000 23.6 vs 19.0) -- the time of
                                             // a small CG iteration is its chain of dependent far reads (every kernel's inputs
                                             // were just written by other CUs), however they are packaged into launches
constexpr int CG2_VW_MAX = CG2_N_MAX / SCSAMD_BLOCK; // virtual update workgroups (vec_grid(n) for n <= CG2_N_MAX)

__global__ __launch_bounds__(SCSAMD_BLOCK) void k_cg2_a(CsrView A, const real *__restrict__ ry, real *tmp, real *x,
                                                       const real *__restrict__ r_old, real *r_new,
                                                       const real *__restrict__ Gp, const real *__restrict__ M,
                                                       const real *__restrict__ p_old, real *p_new_g, int n,
                                                       const real *part_pgp, int cnt_pgp, int gv, CgCtl *ctl, int parity) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cg2_smem[];
  real *pl = reinterpret_cast<real *>(cg2_smem); // n entries: z, then the new p
  __shared__ real prod[NNZ_PER_BLOCK];
  __shared__ real red[SCSAMD_BLOCK / SCSAMD_WAVE];
  __shared__ real wz[CG2_VW_MAX][SCSAMD_BLOCK / SCSAMD_WAVE], wm[CG2_VW_MAX][SCSAMD_BLOCK / SCSAMD_WAVE];
  __shared__ real vpz[CG2_VW_MAX], vpm[CG2_VW_MAX];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int NW = SCSAMD_BLOCK / SCSAMD_WAVE;
  // everything this kernel reads before the product is requested up front
  const int done = ctl->cg_done;
  const real ztr_in = ctl->ztr[parity], tol = ctl->tol;
  real psum = 0;
  for (int i = tid; i < cnt_pgp; i += SCSAMD_BLOCK) psum += part_pgp[i];
  if ((int)blockIdx.x < A.nblk) {
    const int touch = A.rowblk[blockIdx.x + 1] ^ (int)A.blkptr[blockIdx.x + 1];
    asm volatile("" ::"v"(touch));
  }
  if (done) return;
  const real pgp = block_sum(psum, red); // = reduce_partials_sum(part_pgp, cnt_pgp, red) of k_cg_update
  const real alpha = ztr_in / pgp;
  const int nv = n / RVW, gs = gv * SCSAMD_BLOCK;
  const bool writer = blockIdx.x == 0;
  // ---- update (k_cg_update), virtual workgroup by virtual workgroup: lane `tid` of virtual workgroup vw is global lane
  // vw * 256 + tid and owns the vector chunk of that index (gs >= n: one chunk each) -- plus the scalar tail
  // Virtual workgroups are taken four at a time with all their loads requested before anything is used (they are
  // independent; one after the other they would cost a dependent read each).
  constexpr int VB = 4;
  for (int vw0 = 0; vw0 < gv; vw0 += VB) {
    rvec Pq[VB], Gq[VB], Mq[VB], Rq[VB], Xq[VB];
    bool okq[VB];
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int gtid = (vw0 + u) * SCSAMD_BLOCK + tid;
      okq[u] = vw0 + u < gv && gtid < nv;
      Pq[u] = Gq[u] = Mq[u] = Rq[u] = Xq[u] = rvec{};
      if (okq[u]) {
        Pq[u] = ldv(p_old, gtid);
        Gq[u] = ldv(Gp, gtid);
        Mq[u] = ldv(M, gtid);
        Rq[u] = ldv(r_old, gtid);
        if (writer) Xq[u] = ldv(x, gtid);
      }
    }
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int vw = vw0 + u;
      if (vw >= gv) break;
      const int gtid = vw * SCSAMD_BLOCK + tid;
      real ztr = 0, mx = 0;
      if (okq[u]) {
        rvec R = Rq[u], X = Xq[u];
#pragma unroll
        for (int e = 0; e < RVW; ++e) {
          X.v[e] += alpha * Pq[u].v[e];
          const real ri = R.v[e] + (-alpha) * Gq[u].v[e];
          R.v[e] = ri;
          const real zi = ri * Mq[u].v[e];
          pl[gtid * RVW + e] = zi;
          ztr += zi * ri;
          const real a = absval(ri);
          mx = a > mx ? a : mx;
        }
        if (writer) {
          stv(x, gtid, X);
          stv(r_new, gtid, R);
        }
      }
      for (int iv = gtid + gs; iv < nv; iv += gs) { // only when the vector grid is capped below n / 256
        const rvec P = ldv(p_old, iv), G = ldv(Gp, iv), Mv = ldv(M, iv);
        rvec R = ldv(r_old, iv);
#pragma unroll
        for (int e = 0; e < RVW; ++e) {
          const real ri = R.v[e] + (-alpha) * G.v[e];
          R.v[e] = ri;
          const real zi = ri * Mv.v[e];
          pl[iv * RVW + e] = zi;
          ztr += zi * ri;
          const real a = absval(ri);
          mx = a > mx ? a : mx;
        }
        if (writer) {
          rvec X = ldv(x, iv);
#pragma unroll
          for (int e = 0; e < RVW; ++e) X.v[e] += alpha * P.v[e];
          stv(x, iv, X);
          stv(r_new, iv, R);
        }
      }
      for (int i = nv * RVW + gtid; i < n; i += gs) {
        const real pi = p_old[i], gi = Gp[i];
        const real ri = r_old[i] + (-alpha) * gi;
        const real zi = ri * M[i];
        pl[i] = zi;
        ztr += zi * ri;
        const real a = absval(ri);
        mx = a > mx ? a : mx;
        if (writer) {
          x[i] += alpha * pi;
          r_new[i] = ri;
        }
      }
      // block_sum / block_max of the virtual workgroup: wave trees now, the cross-wave step after one barrier
      ztr = wave_sum(ztr);
      mx = wave_max(mx);
      if (lane == 0) {
        wz[vw][wave] = ztr;
        wm[vw][wave] = mx;
      }
    }
  }
  __syncthreads();
  if (tid < gv) {
    real sz = wz[tid][0], sm = wm[tid][0];
    for (int i = 1; i < NW; ++i) {
      sz += wz[tid][i];
      sm = wm[tid][i] > sm ? wm[tid][i] : sm;
    }
    vpz[tid] = sz; // = part_ztr[vw] of k_cg_update
    vpm[tid] = sm; // = part_max[vw]
  }
  __syncthreads();
  // ---- direction (k_cg_direction): same re-reduction of the partials, same tests
  real zs = 0, ms = 0;
  for (int i = tid; i < gv; i += SCSAMD_BLOCK) {
    zs += vpz[i];
    const real v = vpm[i];
    ms = v > ms ? v : ms;
  }
  const real ztr = block_sum(zs, red);
  const real nr = block_max(ms, red);
  const bool conv = nr < tol;
  const bool brk = !conv && ztr_in == (real)0;
  if (!conv && !brk) {
    const real beta = ztr / ztr_in;
    for (int i = tid; i < n; i += SCSAMD_BLOCK) {
      const real pn = pl[i] + beta * p_old[i];
      pl[i] = pn;
      if (writer) p_new_g[i] = pn;
    }
  }
  if (writer && tid == 0) {
    ctl->ztr[parity ^ 1] = ztr;
    ctl->norm_r = nr;
    if (!brk) ctl->iters += 1;
    if (conv || brk || ctl->iters >= ctl->max_its) ctl->cg_done = 1;
  }
  if (conv || brk) return;
  __syncthreads();
  // ---- tmp = R_y^-1 A p, p gathered from LDS
  EpiArgs e{ry, nullptr, nullptr, nullptr};
  real dot = 0;
  csr_stream_blocks<EPI_DIV>(A, pl, tmp, e, prod, red, blockIdx.x, gridDim.x, dot);
}

// (The whole iteration loop in ONE persistent launch of <= 64 co-resident workgroups with grid barriers between the phases was built in
// round 3, bit-identical to the kernel-per-phase path, and measured slower at every size -- 19.9 vs 16.6 us per CG iteration at n = 1000,
// 39.5 vs 21.1 at n = 2e4: an agent-scope release / acquire pair costs more here than a dependent kernel boundary;
// profiles/r3_persistent_pcg.md.)
// ----------------------------------------------------------------------------
// Tiny systems: the WHOLE scs_solve_lin_sys (private.c:284-324) in one launch of one
// 1024-lane workgroup.  Below a few thousand nonzeros a CG iteration is four launches of
// pure latency; one CU does the same work in less time and the per-batch host readback
// disappears because the loop exits on the device.  Same arithmetic as the multi-kernel path
// (row sums in index order; reductions are fixed-order tree reductions).
// ----------------------------------------------------------------------------
constexpr int FUSED_THREADS = 1024;

__device__ __forceinline__ real fused_row(const CsrView &A, const real *x, int r) {
  real acc = 0;
  for (eoff k = A.ptr[r]; k < A.ptr[r + 1]; ++k) acc += A.val[k] * x[A.idx[k]];
  return acc;
}
// y = (R_x + P + A' R_y^-1 A) x ; returns nothing, caller syncs
__device__ void fused_matvec(const CsrView &A, const CsrView &At, const CsrView *P, const real *rx, const real *ry,
                             const real *x, real *tmp, real *y) {
  for (int r = threadIdx.x; r < A.rows; r += FUSED_THREADS) tmp[r] = fused_row(A, x, r) / ry[r];
  __syncthreads();
  for (int r = threadIdx.x; r < At.rows; r += FUSED_THREADS) {
    real acc = P ? fused_row(*P, x, r) : (real)0;
    for (eoff k = At.ptr[r]; k < At.ptr[r + 1]; ++k) acc += At.val[k] * tmp[At.idx[k]];
    y[r] = acc + rx[r] * x[r];
  }
  __syncthreads();
}

__global__ __launch_bounds__(FUSED_THREADS) void k_linsys_fused(CsrView A, CsrView At, CsrView P, int has_P,
                                                                real *b, const real *s, const real *rx,
                                                                const real *ry, const real *M, real *p, real *r,
                                                                real *Gp, real *z, real *tmp, CgCtl *ctl, real tol,
                                                                int form_tol, real tol_scale, long long max_its) {
  __shared__ real red[FUSED_THREADS / SCSAMD_WAVE];
  const int tid = threadIdx.x, n = At.rows, m = A.rows;
  const CsrView *Pp = has_P ? &P : nullptr;
  // ||b||_inf <= 1e-12 short circuit (private.c:296-299)
  real mx = 0;
  for (int i = tid; i < n + m; i += FUSED_THREADS) {
    const real a = absval(b[i]);
    mx = a > mx ? a : mx;
  }
  const real nb = block_max(mx, red);
  real t = tol;
  if (form_tol) { // src/scs.c:745-762 with the warm-start norm taken here
    real mw = 0;
    for (int i = tid; i < n; i += FUSED_THREADS) {
      const real a = absval(s[i]);
      mw = a > mw ? a : mw;
    }
    const real nw = block_max(mw, red) * tol_scale;
    t = t < nw ? t : nw;
    t = (real)0.2 * t;
    t = t > (real)1e-12 ? t : (real)1e-12;
  }
  if (nb <= (real)1e-12) {
    for (int i = tid; i < n + m; i += FUSED_THREADS) b[i] = 0;
    if (tid == 0) {
      ctl->zero_rhs = 1; ctl->cg_done = 1; ctl->iters = 0; ctl->tol = t; ctl->rhs_norm = nb; ctl->norm_r = 0;
    }
    return;
  }
  // b_x += A' R_y^-1 r_y
  for (int i = tid; i < m; i += FUSED_THREADS) tmp[i] = b[n + i] / ry[i];
  __syncthreads();
  for (int j = tid; j < n; j += FUSED_THREADS) {
    real acc = b[j];
    for (eoff k = At.ptr[j]; k < At.ptr[j + 1]; ++k) acc += At.val[k] * tmp[At.idx[k]];
    b[j] = acc;
  }
  __syncthreads();
  // residual of the warm start (private.c:145-160)
  if (s) {
    fused_matvec(A, At, Pp, rx, ry, s, tmp, r);
    for (int i = tid; i < n; i += FUSED_THREADS) {
      r[i] = b[i] - r[i];
      b[i] = s[i];
    }
  } else {
    for (int i = tid; i < n; i += FUSED_THREADS) {
      r[i] = b[i];
      b[i] = 0;
    }
  }
  __syncthreads();
  real ztr = 0, nr = 0;
  for (int i = tid; i < n; i += FUSED_THREADS) {
    const real ri = r[i], zi = ri * M[i], a = absval(ri);
    z[i] = zi;
    ztr += zi * ri;
    nr = a > nr ? a : nr;
  }
  ztr = block_sum(ztr, red);
  nr = block_max(nr, red);
  int iters = 0;
  const real thr = t > (real)1e-12 ? t : (real)1e-12;
  if (!(nr < thr)) {
    for (int i = tid; i < n; i += FUSED_THREADS) p[i] = z[i];
    __syncthreads();
    for (long long it = 0; it < max_its; ++it) {
      fused_matvec(A, At, Pp, rx, ry, p, tmp, Gp);
      real d = 0;
      for (int i = tid; i < n; i += FUSED_THREADS) d += p[i] * Gp[i];
      const real alpha = ztr / block_sum(d, red);
      const real ztr_prev = ztr;
      real zs = 0, ms = 0;
      for (int i = tid; i < n; i += FUSED_THREADS) {
        b[i] += alpha * p[i];
        const real ri = r[i] + (-alpha) * Gp[i];
        r[i] = ri;
        const real zi = ri * M[i], a = absval(ri);
        z[i] = zi;
        zs += zi * ri;
        ms = a > ms ? a : ms;
      }
      ztr = block_sum(zs, red);
      nr = block_max(ms, red);
      if (nr < t) {
        iters = (int)(it + 1);
        break;
      }
      if (ztr_prev == (real)0) {
        iters = (int)it;
        break;
      }
      const real beta = ztr / ztr_prev;
      for (int i = tid; i < n; i += FUSED_THREADS) p[i] = z[i] + beta * p[i];
      __syncthreads();
      iters = (int)(it + 1);
    }
  }
  __syncthreads();
  // y = R_y^-1 (A x - r_y)
  for (int i = tid; i < m; i += FUSED_THREADS) b[n + i] = (-b[n + i] + fused_row(A, b, i)) / ry[i];
  if (tid == 0) {
    ctl->zero_rhs = 0; ctl->cg_done = 1; ctl->iters = iters; ctl->tol = t; ctl->rhs_norm = nb; ctl->norm_r = nr;
    ctl->ztr[0] = ztr; ctl->ztr[1] = 0;
  }
}

// ----------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------
LinSys::~LinSys() {
  if (cg_graph) (void)hipGraphExecDestroy(cg_graph);
  if (own_stream && stream) (void)hipStreamDestroy(stream);
}

// lockstep instantiations (experiment matrix of round 4): waves per workgroup x (barriers per chunk, gather-first pipeline)
template <int E, int WPB, int MODE>
static void launch_lockstep_one(const WaveRowsDev &wd, int g, size_t lds, hipStream_t stream, const WaveView &v, const real *x, real *y,
                                const EpiArgs &e, const int *skip) {
  // many waves' accumulators can pass the 64 KB a kernel gets without asking.  The opt-in is state of the function ON ONE DEVICE and the
  // library serves several devices (and host threads) per process: one flag per device, set under a lock
  static std::atomic<unsigned long long> attr_set[4] = {};
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  const bool tracked = dev >= 0 && dev < 256;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!tracked || !(attr_set[(dev >> 6) & 3].load(std::memory_order_acquire) & bit)) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(csr_wave_lockstep_kernel<E, WPB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  144 * 1024));
    if (tracked) attr_set[(dev >> 6) & 3].fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((csr_wave_lockstep_kernel<E, WPB, MODE>), dim3(g), dim3(WPB * 64), lds, stream, v, x, y, e, skip, wd.accrows);
}
template <int E>
static void launch_lockstep(const WaveRowsDev &wd, int g, size_t lds, hipStream_t stream, const WaveView &v, const real *x, real *y,
                            const EpiArgs &e, const int *skip) {
#define LS_CASE(W, M)                                                                                                  \
  if (wd.ls_wpb == W && wd.ls_bmode == M) return launch_lockstep_one<E, W, M>(wd, g, lds, stream, v, x, y, e, skip);
  LS_CASE(16, 4) LS_CASE(16, 1) LS_CASE(8, 4) LS_CASE(8, 1)
#undef LS_CASE
  throw HipError("scs_amd: lockstep SpMV variant not instantiated");
}

void LinSys::launch_spmv(int epi, const CsrDev &mat, const real *x, real *y, const EpiArgs &e,
                         const int *skip) {
  int slot = -1;
  // one launch in SEVEN is event-timed: the products of a CG iteration alternate A, A', A, A' ..., and a period of 8 (rounds 1-5) kept
  // landing on the same orientation -- the "average launch" was one orientation's (65.3 vs 67.2 us on the headline: close, which is why
  // the rocprofv3 cross-check never flagged it); an odd period alternates
  const bool sample = profiling && ((spmv_sample_ctr++ % 7) == 0);
  if (sample) slot = spmv_timer.start(stream);
  if (mat.wave && mat.wave->built) {
    const WaveRowsDev &wd = *mat.wave;
    const int g = wd.grid();
    const size_t lds = wd.lds_bytes();
    WaveView v = wd.view();
#define WR_LAUNCH(E)                                                                                                   \
  do {                                                                                                                 \
    if (wd.wide) {                                                                                                     \
      hipLaunchKernelGGL((csr_wave_wide_kernel<E>), dim3(g), dim3(WR_BLOCK), lds, stream, v, x, y, e, skip, wd.accrows);  \
    } else if (wd.lockstep) {                                                                                          \
      launch_lockstep<E>(wd, g, lds, stream, v, x, y, e, skip);                                                        \
    } else if (wd.pipelined == 1) hipLaunchKernelGGL((csr_wave_kernel<E, 1>), dim3(g), dim3(WR_BLOCK), lds, stream, v, x, y, e, skip, wd.accrows); \
    else if (wd.pipelined == 2) hipLaunchKernelGGL((csr_wave_kernel<E, 2>), dim3(g), dim3(WR_BLOCK), lds, stream, v, x, y, e, skip, wd.accrows); \
    else hipLaunchKernelGGL((csr_wave_kernel<E, 0>), dim3(g), dim3(WR_BLOCK), lds, stream, v, x, y, e, skip, wd.accrows);                       \
  } while (0)
    switch (epi) {
    case EPI_PLAIN: WR_LAUNCH(EPI_PLAIN); break;
    case EPI_DIV: WR_LAUNCH(EPI_DIV); break;
    case EPI_GP: WR_LAUNCH(EPI_GP); break;
    case EPI_ACC: WR_LAUNCH(EPI_ACC); break;
    case EPI_NEGDIV: WR_LAUNCH(EPI_NEGDIV); break;
    case EPI_GP3: WR_LAUNCH(EPI_GP3); break;
    default: throw HipError("scs_amd: bad spmv epilogue");
    }
#undef WR_LAUNCH
    if (sample) spmv_timer.stop(slot, stream);
    n_spmv++;
    return;
  }
  const int g = mat.grid();
  CsrView v = mat.view();
  switch (epi) {
  case EPI_PLAIN: hipLaunchKernelGGL(csr_stream_kernel<EPI_PLAIN>, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, v, x, y, e, skip); break;
  case EPI_DIV: hipLaunchKernelGGL(csr_stream_kernel<EPI_DIV>, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, v, x, y, e, skip); break;
  case EPI_GP: hipLaunchKernelGGL(csr_stream_kernel<EPI_GP>, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, v, x, y, e, skip); break;
  case EPI_ACC: hipLaunchKernelGGL(csr_stream_kernel<EPI_ACC>, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, v, x, y, e, skip); break;
  case EPI_NEGDIV: hipLaunchKernelGGL(csr_stream_kernel<EPI_NEGDIV>, dim3(g), dim3(SCSAMD_BLOCK), 0, stream, v, x, y, e, skip); break;
  default: throw HipError("scs_amd: bad spmv epilogue");
  }
  if (sample) spmv_timer.stop(slot, stream);
  n_spmv++;
}

static void host_transpose(int rows_out, int cols_out, const eoff *Ap, const int *Ai, const real *Ax, std::vector<eoff> &Cp, std::vector<int> &Ci,
                           std::vector<real> &Cx, bool force_serial = false) {
  host_transpose_t<eoff, real>(rows_out, cols_out, Ap, Ai, Ax, Cp, Ci, Cx, force_serial);
}

void LinSys::init(const CscView *A_csc, const CscView *P_csc, hipStream_t s, CsrPattern *pat) {
  n = A_csc->n;
  m = A_csc->m;
  if (s) {
    stream = s;
    own_stream = false;
  } else {
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    own_stream = true;
  }
  const bool dbg_t = opt_get("debug") != nullptr;
  auto t_now = [] {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  };
  double tp = t_now();
  auto phase = [&](const char *what) {
    if (dbg_t) fprintf(stderr, "[scs_amd linsys init] %-18s %8.1f ms\n", what, t_now() - tp);
    tp = t_now();
  };
  const bool adopt = pat && pat->dev.valid && !pat->rp.empty();
  // CSC(A) is CSR(A'): upload as is -- or adopt the copy the device equilibration left in HBM
  if (adopt) At.adopt(n, m, A_csc->p, pat->dev.cp, pat->dev.ci, pat->dev.cx, stream);
  else At.upload(n, m, A_csc->p, A_csc->i, A_csc->x, stream);
  phase(adopt ? "adopt At" : "upload At");
  if (WaveRowsDev::wanted(m, A_csc->p, n)) {
    At.wave = new WaveRowsDev();
    wave_build(*At.wave, n, m, A_csc->p, A_csc->i, A_csc->x, At, stream);
    phase("wave-rows At");
    if (dbg_t) fprintf(stderr, "[scs_amd linsys init] A' gathers: %.3f distinct lines per entry -> %s stream; layout built on the %s\n", At.wave->lines_per_entry, At.wave->pipelined ? "pipelined" : "plain", At.wave->built_on_device ? "device" : "host");
  }
  if (adopt) { // CSR(A) is in HBM too; its host column indices / values are only fetched if the host layout builder needs them
    A.adopt(m, n, pat->rp.data(), pat->dev.rp, pat->dev.rj, pat->dev.rx, stream);
    pat->dev.valid = false;
    phase("adopt A");
    if (WaveRowsDev::wanted(n, pat->rp.data(), m)) {
      A.wave = new WaveRowsDev();
      wave_build(*A.wave, m, n, pat->rp.data(), pat->rj.empty() ? nullptr : pat->rj.data(), nullptr, A, stream);
      phase("wave-rows A");
      if (dbg_t) fprintf(stderr, "[scs_amd linsys init] A  gathers: %.3f distinct lines per entry -> %s stream; layout built on the %s\n", A.wave->lines_per_entry, A.wave->pipelined ? "pipelined" : "plain", A.wave->built_on_device ? "device" : "host");
    }
  } else {
    std::vector<eoff> Cp_own;
    std::vector<int> Ci_own;
    std::vector<real> Cx;
    const bool have_pat = pat && !pat->empty() && !pat->rj.empty() && !pat->pos.empty();
    if (have_pat) { // values follow the cached pattern: a gather, not a sort
      const size_t nz = (size_t)A_csc->p[n];
      Cx.resize(nz);
      for (size_t q = 0; q < nz; ++q) Cx[q] = A_csc->x[pat->pos[q]];
    } else {
      host_transpose(m, n, A_csc->p, A_csc->i, A_csc->x, Cp_own, Ci_own, Cx);
    }
    const std::vector<eoff> &Cp = have_pat ? pat->rp : Cp_own;
    const std::vector<int> &Ci = have_pat ? pat->rj : Ci_own;
    phase("transpose");
    A.upload(m, n, Cp.data(), Ci.data(), Cx.data(), stream);
    phase("upload A");
    if (WaveRowsDev::wanted(n, Cp.data(), m)) {
      A.wave = new WaveRowsDev();
      wave_build(*A.wave, m, n, Cp.data(), Ci.data(), Cx.data(), A, stream);
      phase("wave-rows A");
      if (dbg_t) fprintf(stderr, "[scs_amd linsys init] A  gathers: %.3f distinct lines per entry -> %s stream; layout built on the %s\n", A.wave->lines_per_entry, A.wave->pipelined ? "pipelined" : "plain", A.wave->built_on_device ? "device" : "host");
    }
  }
  has_P = P_csc != nullptr;
  if (has_P) {
    // expand the stored upper triangle to the full symmetric matrix (CSR == CSC)
    const eoff *Pp_ = P_csc->p;
    const int *Pi = P_csc->i;
    const real *Px = P_csc->x;
    std::vector<eoff> cnt((size_t)n + 1, 0);
    std::vector<real> pd((size_t)n, 0);
    for (int j = 0; j < n; ++j)
      for (eoff k = Pp_[j]; k < Pp_[j + 1]; ++k) {
        const int i = Pi[k];
        cnt[(size_t)j + 1]++;
        if (i != j) cnt[(size_t)i + 1]++;
        else pd[j] += Px[k]; // private.c:69-75
      }
    for (int i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    std::vector<eoff> nxt(cnt.begin(), cnt.end() - 1);
    std::vector<int> Fi((size_t)cnt[n]);
    std::vector<real> Fx((size_t)cnt[n]);
    for (int j = 0; j < n; ++j)
      for (eoff k = Pp_[j]; k < Pp_[j + 1]; ++k) {
        const int i = Pi[k];
        eoff q = nxt[j]++; // row j, column i
        Fi[q] = i;
        Fx[q] = Px[k];
        if (i != j) {
          q = nxt[i]++; // row i, column j
          Fi[q] = j;
          Fx[q] = Px[k];
        }
      }
    P.upload(n, n, cnt.data(), Fi.data(), Fx.data(), stream);
    Pdiag.alloc(n);
    Pdiag.upload(pd.data(), n, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    Pp.alloc(n);
  }
  {
    // one-workgroup path for small systems (SCS_AMD_FUSED=0/1 forces either path in tests)
    const long long nnzA = A_csc->p[n];
    // measured: at n=1000/nnz=32000 one CU is already 3x slower than the multi-kernel path, so
    // this is for genuinely tiny systems only, where a CG iteration is pure launch latency
    use_fused = nnzA <= 4096 && n <= 1024 && m <= 4096;
    if (const char *e = opt_get("fused")) use_fused = atoi(e) != 0;
    // below this size a CG kernel is shorter than the host's cost of launching it: replay the
    // iterations from a captured graph instead (above it launches are hidden behind the kernels)
    // x, r and M are streamed once per CG iteration and nothing gathers from them: with the non-temporal policy they stop pushing
    // the matrix streams and the gathered vectors out of L2 / the Infinity Cache -- measured at the headline size 171.3 -> 169.7 us
    // per CG iteration (same bits); only where the matrices do not fit the Infinity Cache anyway.  SCS_AMD_VEC_NT = 0 | 1 | 3 (all
    // streams of the kernel: no better) forces a mode.
    nt_mode = nnzA >= 4000000 ? 1 : 0;
    if (const char *e = opt_get("vec_nt")) nt_mode = atoi(e);
    if (const char *e = opt_get("dir_mode")) dir_mode = atoi(e);
    use_graph = nnzA <= CG_GRAPH_MAX_NNZ;
    if (const char *e = opt_get("graph")) use_graph = atoi(e) != 0;
    // two launches per CG iteration (k_cg2_a + the transposed product): n small enough for p in LDS, no P
    // (k_cg2_a runs the A product through csr_stream_blocks and sums At.grid() partials: not with the wave-owned-rows
    // layout, whose GP product leaves a different partial count -- very tall A with n <= 1024 takes the four-kernel path)
    const bool no_wave = !(A.wave && A.wave->built) && !(At.wave && At.wave->built);
    use_cg2 = !use_fused && !has_P && no_wave && n <= CG2_N_MAX && n >= 2 * RVW;
    if (const char *e = opt_get("cg2")) use_cg2 = atoi(e) != 0 && !has_P && no_wave && n <= CG2_N_MAX && n >= 2 * RVW;
    // three launches per CG iteration (k_cg3_update), SCS_AMD_CG3=1: possible wherever the transposed product runs through the wave-owned-
    // rows kernels (their epilogue carries the two extra dot products)
    // (fp64 only: in fp32 the expansion behind beta loses ~eps x 10..100 relative, which is the size of fp32 CG's own rounding -- not worth
    // a 2 % shorter iteration there)
    const bool cg3_ok = sizeof(real) == 8 && !use_fused && !use_cg2 && At.wave && At.wave->built && vec_grid(n) <= PART_CAP / 4;
    // OFF by default: measured at the headline size it buys 1.1 us of a 159.6 us CG iteration (profiles/r5_cg_vector_kernels.md: the launch
    // and the re-reduction it removes are paid back by two more vector reads in the product's epilogue and one product enqueued past
    // convergence per solve) and it changes the rounding of every iterate -- not worth leaving the reference's recurrence for.
    use_cg3 = false;
    if (const char *e = opt_get("cg3")) use_cg3 = atoi(e) != 0 && cg3_ok;
    if (use_cg3) use_graph = false;
    if (use_cg2) {
      p2.alloc(n);
      r2.alloc(n);
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cg2_a), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(CG2_N_MAX * sizeof(real))));
    }
  }
  rx.alloc(n);
  ry.alloc(m);
  M.alloc(n);
  p.alloc(n);
  r.alloc(n);
  Gp.alloc(n);
  z.alloc(n);
  tmp.alloc(m);
  // 64-bit entry positions exercised without a 26 GB matrix (DLONG build only, tests/test_dlong_gpu.py): every stored entry position
  // (row pointers, row-block first entries, unit entry ranges) gets +bias and the arrays they index are handed to the kernels shifted by
  // -bias -- a kernel that narrows a position to 32 bits anywhere between the table and the load reads the wrong entry.  bias must be a
  // multiple of 4 (the wave kernels load 16 bytes of 4-byte words at a time).
  if (sizeof(eoff) == 8)
    if (const char *e = opt_get("test_offset_bias")) {
      const long long bias = atoll(e) & ~3LL;
      auto shift = [&](CsrDev &M) {
        if (M.rows <= 0 || bias == 0) return;
        hipLaunchKernelGGL(k_add_bias, dim3(vec_grid(M.rows + 1)), dim3(SCSAMD_BLOCK), 0, stream, M.ptr.p, (long long)M.rows + 1, (eoff)bias);
        hipLaunchKernelGGL(k_add_bias, dim3(vec_grid(M.nblk + 1)), dim3(SCSAMD_BLOCK), 0, stream, M.blkptr.p, (long long)M.nblk + 1, (eoff)bias);
        M.bias = bias;
        if (M.wave && M.wave->built) {
          hipLaunchKernelGGL(k_add_bias, dim3(vec_grid(2LL * M.wave->nunit)), dim3(SCSAMD_BLOCK), 0, stream, M.wave->useg.p, 2LL * M.wave->nunit, (eoff)bias);
          M.wave->bias = bias;
        }
      };
      shift(A);
      shift(At);
      if (has_P) shift(P);
      HIP_CHECK(hipGetLastError());
    }
  partA.alloc(PART_CAP);
  partB.alloc(PART_CAP);
  if (use_cg3) {
    partC.alloc(PART_CAP);
    partD.alloc(PART_CAP);
  }
  ctl.alloc(1);
  hctl.alloc(1);
  HIP_CHECK(hipStreamSynchronize(stream));
}

void LinSys::build_preconditioner() {
  hipLaunchKernelGGL(k_precond, dim3(vec_grid(n)), dim3(SCSAMD_BLOCK), 0, stream, At.view(), rx.p, ry.p,
                     has_P ? Pdiag.p : (const real *)nullptr, M.p);
  if (shard) { // diag(G) = sum over the slabs of (R_x / N + diag(A_r' R_r^-1 A_r)): sum the reciprocals of the local M
    hipLaunchKernelGGL(k_reciprocal, dim3(vec_grid(n)), dim3(SCSAMD_BLOCK), 0, stream, M.p, n);
    shard_allreduce(M.p, (size_t)n, 0);
    hipLaunchKernelGGL(k_reciprocal, dim3(vec_grid(n)), dim3(SCSAMD_BLOCK), 0, stream, M.p, n);
  }
}

void LinSys::set_shard(ShardHook *h) {
  shard = h;
  if (h) { // the small-system shortcuts assume the whole operator is local
    if (has_P) throw HipError("scs_amd: a row-sharded system with P is not supported");
    use_fused = false;
    use_cg2 = false;
    use_cg3 = false;
    use_graph = false;
  }
}

void LinSys::shard_allreduce(real *buf, size_t count, int op) {
  int slot = -1;
  if (profiling && (n_allreduce & 3) == 0) slot = ar_timer.start(stream);
  if (!shard || !shard->allreduce || shard->allreduce(shard->ctx, buf, count, op, stream) != 0)
    throw HipError("scs_amd: all-reduce of the row-sharded solve failed");
  if (slot >= 0) ar_timer.stop(slot, stream);
  ++n_allreduce;
}

void LinSys::set_diag_r_host(const real *diag_r) {
  if (dr_stage.n < (size_t)(n + m)) dr_stage.alloc((size_t)n + m);
  dr_stage.upload(diag_r, (size_t)n + m, stream);
  HIP_CHECK(hipStreamSynchronize(stream)); // caller's buffer may be pageable and reused
  set_diag_r_dev(dr_stage.p);
}

void LinSys::set_diag_r_dev(const real *d) {
  HIP_CHECK(hipMemcpyAsync(rx.p, d, (size_t)n * sizeof(real), hipMemcpyDeviceToDevice, stream));
  HIP_CHECK(hipMemcpyAsync(ry.p, d + n, (size_t)m * sizeof(real), hipMemcpyDeviceToDevice, stream));
  build_preconditioner();
}

void LinSys::mat_vec_dev(const real *x, real *y_out, real *dot_partials) {
  EpiArgs e1{ry.p, nullptr, nullptr, nullptr};
  launch_spmv(EPI_DIV, A, x, tmp.p, e1, nullptr);
  if (has_P) {
    EpiArgs ep{nullptr, nullptr, nullptr, nullptr};
    launch_spmv(EPI_PLAIN, P, x, Pp.p, ep, nullptr);
  }
  EpiArgs e2{rx.p, x, has_P ? Pp.p : nullptr, dot_partials};
  launch_spmv(EPI_GP, At, tmp.p, y_out, e2, nullptr);
  n_matvecs++;
}

void LinSys::mul_A(const real *x_n, real *y_m) {
  EpiArgs e{nullptr, nullptr, nullptr, nullptr};
  launch_spmv(EPI_PLAIN, A, x_n, y_m, e, nullptr);
}
void LinSys::mul_At(const real *y_m, real *x_n) {
  EpiArgs e{nullptr, nullptr, nullptr, nullptr};
  launch_spmv(EPI_PLAIN, At, y_m, x_n, e, nullptr);
}
void LinSys::mul_P(const real *x_n, real *y_n) {
  if (!has_P) {
    HIP_CHECK(hipMemsetAsync(y_n, 0, (size_t)n * sizeof(real), stream));
    return;
  }
  EpiArgs e{nullptr, nullptr, nullptr, nullptr};
  launch_spmv(EPI_PLAIN, P, x_n, y_n, e, nullptr);
}

void LinSys::harvest_timers() {
  spmv_timer.harvest();
  cg_timer.harvest();
}


// one PCG iteration = K1 (z = R_y^-1 A p), [P p], K2 (Gp, partial p'Gp), K3 (alpha, x, r, z, partial
// z'r, |r|), K4 (stop test, beta, p).  K3 and K4 stay two launches: fused behind a grid barrier (p and z in registers
// across it, z never written) they were bit-identical and SLOWER -- 184 vs 172 us per CG iteration at n = 1e6, 48 vs 39
// at n = 2e5 in the solver, vector part 33.7 vs 21.8 us in lab/cgfuse_lab.hip (profiles/r3_cgfuse_lab.md); `b` is where the solve keeps x: always b_stage / the caller's
// device vector, fixed per LinSys user, so it is passed through cg_x
// small systems: iteration `it` >= 1 = k_cg2_a (update + direction of iteration it-1, then z = R_y^-1 A p_it) and the
// transposed product; p_j lives in pbuf[j & 1].  Iteration 0 (no update yet) uses the plain A product.
void LinSys::enqueue_cg2_iteration(long long it) {
  CgCtl *c = ctl.p;
  real *pbuf[2] = {p.p, p2.p}, *rbuf[2] = {r.p, r2.p};
  const int gv = vec_grid(n);
  const int gAt = (At.wave && At.wave->built) ? At.wave->grid() : At.grid(); // partials the GP product really wrote
  real *p_cur = pbuf[it & 1];
  if (it == 0) {
    EpiArgs e1{ry.p, nullptr, nullptr, nullptr};
    launch_spmv(EPI_DIV, A, p_cur, tmp.p, e1, &c->cg_done);
  } else {
    const int q = (int)((it - 1) & 1);
    hipLaunchKernelGGL(k_cg2_a, dim3(A.grid()), dim3(SCSAMD_BLOCK), (size_t)n * sizeof(real), stream, A.view(), ry.p, tmp.p, cg_x,
                       rbuf[(it - 1) & 1], rbuf[it & 1], Gp.p, M.p, pbuf[(it - 1) & 1], p_cur, n, partA.p, gAt, gv, c, q);
    n_spmv++;
  }
  EpiArgs e2{rx.p, p_cur, nullptr, partA.p};
  launch_spmv(EPI_GP, At, tmp.p, Gp.p, e2, &c->cg_done);
}

void LinSys::enqueue_cg_iteration(int q) {
  CgCtl *c = ctl.p;
  const int gv = vec_grid(n);
  const int gAt = (At.wave && At.wave->built) ? At.wave->grid() : At.grid();
  real *part_pgp = partA.p, *part_ztr = partB.p, *part_max = partB.p + PART_CAP / 2;
  EpiArgs e1{ry.p, nullptr, nullptr, nullptr};
  launch_spmv(EPI_DIV, A, p.p, tmp.p, e1, &c->cg_done);
  if (has_P) {
    EpiArgs ep{nullptr, nullptr, nullptr, nullptr};
    launch_spmv(EPI_PLAIN, P, p.p, Pp.p, ep, &c->cg_done);
  }
  EpiArgs e2{rx.p, p.p, has_P ? Pp.p : nullptr, shard ? (real *)nullptr : part_pgp};
  launch_spmv(EPI_GP, At, tmp.p, Gp.p, e2, &c->cg_done);
  int cnt_pgp = gAt;
  if (shard) { // Gp holds this slab's term: sum over the ranks, then the dot product (iterations past convergence still take part in
               // the collective -- every rank enqueues the same sequence -- and move an untouched Gp)
    shard_allreduce(Gp.p, (size_t)n, 0);
    hipLaunchKernelGGL(k_dot_partial, dim3(gv), dim3(SCSAMD_BLOCK), 0, stream, (const real *)p.p, (const real *)Gp.p, n, part_pgp, &c->cg_done);
    cnt_pgp = gv;
  }
  hipLaunchKernelGGL(k_cg_update, dim3(gv), dim3(SCSAMD_BLOCK), 0, stream, cg_x, r.p, z.p, p.p, Gp.p, M.p, n,
                     part_pgp, cnt_pgp, part_ztr, part_max, c, q, nt_mode);
  hipLaunchKernelGGL(k_cg_direction, dim3(gv), dim3(SCSAMD_BLOCK), 0, stream, p.p, z.p, n, part_ztr, part_max,
                     gv, c, q, dir_mode);
}

// three-kernel iteration (k_cg3_update): K1, [P p], K2 with the three dot products in its epilogue, the fused vector kernel
void LinSys::enqueue_cg3_iteration(long long it) {
  CgCtl *c = ctl.p;
  const int gv = vec_grid(n);
  const int gAt = At.wave->grid();
  const int q = (int)(it & 1);
  real *ztr_cur = partB.p + q * (PART_CAP / 4), *max_cur = partB.p + PART_CAP / 2 + q * (PART_CAP / 4);
  real *ztr_prev = partB.p + (q ^ 1) * (PART_CAP / 4), *max_prev = partB.p + PART_CAP / 2 + (q ^ 1) * (PART_CAP / 4);
  EpiArgs e1{ry.p, nullptr, nullptr, nullptr};
  launch_spmv(EPI_DIV, A, p.p, tmp.p, e1, &c->cg_done);
  if (has_P) {
    EpiArgs ep{nullptr, nullptr, nullptr, nullptr};
    launch_spmv(EPI_PLAIN, P, p.p, Pp.p, ep, &c->cg_done);
  }
  EpiArgs e2{rx.p, p.p, has_P ? Pp.p : nullptr, partA.p};
  e2.mv = M.p;
  e2.rv = r.p;
  e2.partial2 = partC.p;
  e2.partial3 = partD.p;
  launch_spmv(EPI_GP3, At, tmp.p, Gp.p, e2, &c->cg_done);
  hipLaunchKernelGGL(k_cg3_update, dim3(gv), dim3(SCSAMD_BLOCK), 0, stream, cg_x, r.p, p.p, Gp.p, M.p, n, partA.p, partC.p, partD.p, gAt,
                     ztr_prev, max_prev, gv, ztr_cur, max_cur, c, (int)std::min<long long>(it, 2147483647LL));
}

// Capture CG_GRAPH_ITERS iterations into an executable graph.  Any failure leaves cg_graph null
// and the loop keeps launching kernels one by one (same kernels, same order).
bool LinSys::build_cg_graph() {
  cg_graph_tried = true;
  hipGraph_t g = nullptr;
  // relaxed: other host threads (concurrent solves, each on its own stream) keep allocating and
  // copying while this thread captures; nothing they do touches this stream
  if (hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  bool ok = true;
  try {
    for (int j = 0; j < CG_GRAPH_ITERS; ++j) {
      if (use_cg2) enqueue_cg2_iteration(1 + j); // replays always start at an odd iteration: same parity pattern
      else enqueue_cg_iteration(j & 1);
    }
  } catch (...) {
    ok = false;
  }
  if (hipStreamEndCapture(stream, &g) != hipSuccess || !g) ok = false;
  if (ok && hipGraphInstantiate(&cg_graph, g, nullptr, nullptr, 0) != hipSuccess) {
    cg_graph = nullptr;
    ok = false;
  }
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  return ok;
}

int LinSys::solve_dev(real *b, const real *s, real tol, const real *warm_part, int warm_cnt,
                      real warm_scale) {
  const int gv = vec_grid(n), gnm = vec_grid((long long)n + m);
  CgCtl *c = ctl.p;
  static const bool debug = opt_get("debug") != nullptr;
  if (cg_x != b) { // the captured graph bakes the solution vector's address in
    if (cg_graph) (void)hipGraphExecDestroy(cg_graph);
    cg_graph = nullptr;
    cg_graph_tried = false;
    cg_x = b;
  }
  int cg_slot = -1;
  if (profiling) cg_slot = cg_timer.start(stream);

  if (use_fused) { // small system: one launch, the loop exits on the device
    // B2 passes the warm start as `s` and asks for tol = max(1e-12, 0.2 min(tol, |s|_inf * warm_scale))
    hipLaunchKernelGGL(k_linsys_fused, dim3(1), dim3(FUSED_THREADS), 0, stream, A.view(), At.view(),
                       has_P ? P.view() : A.view(), has_P ? 1 : 0, b, s, rx.p, ry.p, M.p, p.p, r.p, Gp.p, z.p, tmp.p,
                       c, tol, warm_part ? 1 : 0, warm_scale, 10LL * n);
    HIP_CHECK(hipMemcpyAsync(hctl.p, c, sizeof(CgCtl), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipGetLastError());
    if (profiling) cg_timer.stop(cg_slot, stream);
    const int fits = hctl.p->iters;
    if (debug)
      fprintf(stderr, "[scs_amd pcg fused] iters=%d zero=%d |r|=%.3e tol=%.3e |b|=%.3e\n", fits, hctl.p->zero_rhs,
              (double)hctl.p->norm_r, (double)hctl.p->tol, (double)hctl.p->rhs_norm);
    n_matvecs += fits + (s ? 1 : 0);
    last_its = fits;
    tot_cg_its += fits;
    n_solves++;
    return fits;
  }

  int rhs_cnt = gnm;
  if (shard) { // |[r_x; r_y]|_inf over ALL slabs: element-wise maximum of the (zero-padded) partial arrays
    rhs_cnt = PART_CAP / 2; // >= any vec_grid
    HIP_CHECK(hipMemsetAsync(partA.p, 0, (size_t)rhs_cnt * sizeof(real), stream));
  }
  hipLaunchKernelGGL(k_absmax_partial, dim3(gnm), dim3(SCSAMD_BLOCK), 0, stream, b, n + m, partA.p);
  if (shard) shard_allreduce(partA.p, (size_t)rhs_cnt, 1);
  hipLaunchKernelGGL(k_rhs_prep, dim3(gnm), dim3(SCSAMD_BLOCK), 0, stream, b, ry.p, tmp.p, n, m, partA.p,
                     rhs_cnt, c, tol, warm_part, warm_cnt, warm_scale,
                     (int)std::min<long long>(10LL * n, 2147483647LL));
  // b_x += A' R_y^-1 r_y   (private.c:305); sharded: r_x counts once (rank 0 keeps it), every slab adds its part, then the sum
  if (shard) hipLaunchKernelGGL(k_zero_unless_rank0, dim3(gv), dim3(SCSAMD_BLOCK), 0, stream, b, n, shard->rank, (const int *)&c->zero_rhs);
  {
    EpiArgs e{nullptr, nullptr, nullptr, nullptr};
    launch_spmv(EPI_ACC, At, tmp.p, b, e, &c->zero_rhs);
  }
  if (shard) shard_allreduce(b, (size_t)n, 0);
  if (s) { // r = G s  (private.c:153)
    EpiArgs e1{ry.p, nullptr, nullptr, nullptr};
    launch_spmv(EPI_DIV, A, s, tmp.p, e1, &c->zero_rhs);
    if (has_P) {
      EpiArgs ep{nullptr, nullptr, nullptr, nullptr};
      launch_spmv(EPI_PLAIN, P, s, Pp.p, ep, &c->zero_rhs);
    }
    EpiArgs e2{rx.p, s, has_P ? Pp.p : nullptr, nullptr};
    launch_spmv(EPI_GP, At, tmp.p, r.p, e2, &c->zero_rhs);
    if (shard) shard_allreduce(r.p, (size_t)n, 0);
    n_matvecs++;
  }
  hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(SCSAMD_BLOCK), 0, stream, b, s, r.p, z.p, M.p, n, partA.p,
                     partB.p, c);
  hipLaunchKernelGGL(k_cg_start, dim3(gv), dim3(SCSAMD_BLOCK), 0, stream, p.p, z.p, n, partA.p, partB.p, gv, c);

  // ---- iteration batches ---------------------------------------------------
  const long long max_its = 10LL * n; // private.c:307
  long long it = 0;
  int batch = std::max(4, std::min(last_its + 1, 4096));
  // partial arrays: partA <- p'Gp (K2), partB <- z'r and partB+PART_CAP/2 <- |r| (K3)
  if (use_cg2) { // iteration 0 has no update to fold in; everything after it is k_cg2_a + the transposed product
    enqueue_cg2_iteration(0);
    it = 1;
  }
  if (use_graph && !profiling && !cg_graph_tried) build_cg_graph();
  for (;;) {
    // cg2: the update/direction of iteration j runs inside iteration j+1's first kernel, so one more is enqueued
    const int extra = (use_cg2 || use_cg3) ? 1 : 0; // the launch that evaluates the last iteration's stop test
    int nb = (int)std::min<long long>(batch + (use_cg3 ? 1 : 0), max_its + extra - it);
    if (nb < 1) nb = 1;
    if (cg_graph && !profiling) {
      // whole graphs only (the parity of the double-buffered z'r slot restarts with each graph);
      // iterations enqueued past convergence are no-ops, as with individual launches
      const int ng = (nb + CG_GRAPH_ITERS - 1) / CG_GRAPH_ITERS;
      for (int g = 0; g < ng; ++g) HIP_CHECK(hipGraphLaunch(cg_graph, stream));
      n_graph_launches += ng;
      nb = ng * CG_GRAPH_ITERS;
    } else {
      for (int j = 0; j < nb; ++j) {
        if (use_cg2) enqueue_cg2_iteration(it + j);
        else if (use_cg3) enqueue_cg3_iteration(it + j);
        else enqueue_cg_iteration((int)((it + j) & 1));
      }
    }
    it += nb;
    HIP_CHECK(hipMemcpyAsync(hctl.p, c, sizeof(CgCtl), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (debug)
      fprintf(stderr, "[scs_amd pcg] enq=%lld iters=%d done=%d zero=%d |r|=%.3e tol=%.3e ztr=(%.3e,%.3e) |b|=%.3e\n",
              it, hctl.p->iters, hctl.p->cg_done, hctl.p->zero_rhs, (double)hctl.p->norm_r,
              (double)hctl.p->tol, (double)hctl.p->ztr[0], (double)hctl.p->ztr[1], (double)hctl.p->rhs_norm);
    if (hctl.p->cg_done || it >= max_its + extra) break;
    batch = std::max(4, std::min(last_its / 4 + 1, 1024));
  }
  const int its = hctl.p->iters;
  n_matvecs += its;
  // y = R_y^-1 (A x - r_y)   (private.c:313-317)
  {
    EpiArgs e{ry.p, nullptr, nullptr, nullptr};
    launch_spmv(EPI_NEGDIV, A, b, b + n, e, &c->zero_rhs);
  }
  if (profiling) cg_timer.stop(cg_slot, stream);
  HIP_CHECK(hipGetLastError());
  if (const char *tf = opt_get("trace_file")) { // same record as oracle/trace_linsys.c
    real x0 = 0;
    HIP_CHECK(hipMemcpyAsync(&x0, b, sizeof(real), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (FILE *f = fopen(tf, n_solves == 0 ? "w" : "a")) {
      fprintf(f, "%lld tol=%.17g nb=%.17g its=%d x0=%.17g\n", n_solves, (double)hctl.p->tol,
              (double)hctl.p->rhs_norm, its, (double)x0);
      fclose(f);
    }
  }
  last_its = its;
  tot_cg_its += its;
  n_solves++;
  return its;
}

} // namespace scsamd

// ============================================================================
// B1: the reference's linear-system plugin ABI (include/linsys.h:25-71)
// ============================================================================
using namespace scsamd;

struct SCS_LIN_SYS_WORK {
  LinSys ls;
  int device = 0;
};

// Device selection is per host thread (like HIP's own current device); a thread that never chose one
// inherits the most recent choice of any thread.  Workspaces remember the device they were created on
// and every entry point that takes one makes it current first.
#include <atomic>
static std::atomic<int> g_default_device{0};
static thread_local int t_device = -1;
namespace scsamd {
int selected_device() { return t_device >= 0 ? t_device : g_default_device.load(std::memory_order_relaxed); }
}

extern "C" {

scs_int scs_amd_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return -1;
  return c;
}

scs_int scs_amd_set_device(scs_int dev) {
  if (hipSetDevice(dev) != hipSuccess) return -1;
  t_device = dev;
  g_default_device.store(dev, std::memory_order_relaxed);
  return 0;
}

// free device memory in bytes on the current device, < 0 on failure (tests of the failure convention: "nothing leaked")
long long scs_amd_device_free_bytes(void) {
  size_t fr = 0, tot = 0;
  if (hipSetDevice(selected_device()) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return -1;
  return (long long)fr;
}

// test hook: arm (k >= 1) or disarm (k <= 0) the fault injection of common.h's hip_check; returns the previous countdown
long long scs_amd_test_fail_at(long long k) { return fail_countdown().exchange(k > 0 ? k : 0); }

// the options entry (options.h): process-wide, read when a workspace is created
scs_int scs_amd_set_option(const char *key, const char *value) { return (scs_int)opt_set(key, value); }
const char *scs_amd_get_option(const char *key) { return key ? opt_get(key) : nullptr; }
// the table itself, one row per line "key<TAB>class<TAB>numerics<TAB>values<TAB>meaning": what INTEGRATION.md section 5 prints and
// tests/test_options.py checks.  Returns the length needed (excluding the terminating zero); writes at most cap bytes.
scs_int scs_amd_list_options(char *buf, scs_int cap) {
  static const char *cls[] = {"supported", "ab", "test", "diag"};
  int n;
  const OptRow *r = opt_rows(&n);
  std::string out;
  for (int i = 0; i < n; ++i) {
    out += r[i].key;
    out += '\t';
    out += cls[r[i].cls];
    out += '\t';
    out += (char)('0' + r[i].numerics);
    out += '\t';
    out += r[i].values;
    out += '\t';
    out += r[i].doc;
    out += '\n';
  }
  if (buf && cap > 0) {
    const size_t c = std::min(out.size(), (size_t)cap - 1);
    memcpy(buf, out.data(), c);
    buf[c] = 0;
  }
  return (scs_int)out.size();
}

const char *scs_get_lin_sys_method(void) { return "sparse-indirect-pcg-hip-gfx950"; }

ScsLinSysWork *scs_init_lin_sys_work(const ScsMatrix *A, const ScsMatrix *P, const scs_float *diag_r) {
  if (!A || !diag_r) return nullptr;
  ScsLinSysWork *w = nullptr;
  try {
    const int dev = selected_device(); // snapshot once: another thread may re-select concurrently
    HIP_CHECK(hipSetDevice(dev));
    if (!fits_int32(A) || !fits_int32(P)) throw HipError("scs_amd: matrix sizes / nonzero count exceed 32-bit device indexing");
    w = new ScsLinSysWork();
    w->device = dev;
    {
      CscArg a(A), pm(P); // aliases the caller's arrays; under -DDLONG a narrowed copy, dropped after init
      w->ls.init(a.ptr(), pm.ptr(), nullptr);
    }
    w->ls.b_stage.alloc((size_t)A->n + A->m);
    w->ls.s_stage.alloc((size_t)A->n);
    w->ls.set_diag_r_host(diag_r);
    HIP_CHECK(hipStreamSynchronize(w->ls.stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    delete w;
    return nullptr;
  }
  return w;
}

scs_int scs_solve_lin_sys(ScsLinSysWork *w, scs_float *b, const scs_float *s, scs_float tol) {
  if (!w || !b) return -1;
  if (tol <= 0.) {
    // same diagnostic as private.c:288-292: caller was not built with -DINDIRECT=1
    printf("Warning: tol = %4f <= 0, likely compiled without setting INDIRECT flag.\n", (double)tol);
  }
  try {
    HIP_CHECK(hipSetDevice(w->device));
    LinSys &ls = w->ls;
    const size_t n = ls.n, m = ls.m;
    ls.b_stage.upload(b, n + m, ls.stream);
    if (s) ls.s_stage.upload(s, n, ls.stream);
    ls.solve_dev(ls.b_stage.p, s ? ls.s_stage.p : nullptr, tol);
    ls.b_stage.download(b, n + m, ls.stream);
    HIP_CHECK(hipStreamSynchronize(ls.stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return 0;
}

scs_int scs_update_lin_sys_diag_r(ScsLinSysWork *w, const scs_float *new_diag_r) {
  if (!w || !new_diag_r) return -1;
  try {
    HIP_CHECK(hipSetDevice(w->device));
    w->ls.set_diag_r_host(new_diag_r);
    HIP_CHECK(hipStreamSynchronize(w->ls.stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return 0;
}

void scs_free_lin_sys_work(ScsLinSysWork *w) {
  if (!w) return;
  (void)hipSetDevice(w->device);
  delete w;
}

// operator pieces on device pointers (row-sharded systems, scs_amd/shard.py)
static scs_int with_work(ScsLinSysWork *w, const void *a, const void *b, void (*fn)(LinSys &, const real *, real *)) {
  if (!w || !a || !b) return -1;
  try {
    HIP_CHECK(hipSetDevice(w->device));
    fn(w->ls, static_cast<const real *>(a), static_cast<real *>(const_cast<void *>(b)));
    HIP_CHECK(hipGetLastError());
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return 0;
}
scs_int scs_amd_linsys_mat_vec_dev(ScsLinSysWork *w, const scs_float *x_dev, scs_float *y_dev) {
  return with_work(w, x_dev, y_dev, [](LinSys &ls, const real *x, real *y) { ls.mat_vec_dev(x, y, nullptr); });
}
scs_int scs_amd_linsys_mul_a_dev(ScsLinSysWork *w, const scs_float *x_dev, scs_float *y_dev) {
  return with_work(w, x_dev, y_dev, [](LinSys &ls, const real *x, real *y) { ls.mul_A(x, y); });
}
scs_int scs_amd_linsys_mul_at_dev(ScsLinSysWork *w, const scs_float *y_dev, scs_float *x_dev) {
  return with_work(w, y_dev, x_dev, [](LinSys &ls, const real *y, real *x) { ls.mul_At(y, x); });
}
scs_int scs_amd_linsys_sync(ScsLinSysWork *w) {
  if (!w) return -1;
  if (hipSetDevice(w->device) != hipSuccess) return -1;
  return hipStreamSynchronize(w->ls.stream) == hipSuccess ? 0 : -1;
}

void scs_amd_linsys_set_profiling(ScsLinSysWork *w, scs_int on) {
  if (w) w->ls.profiling = on != 0;
}

void scs_amd_linsys_get_stats(const ScsLinSysWork *w, ScsAmdStats *out) {
  if (!w || !out) return;
  LinSys &ls = const_cast<LinSys &>(w->ls);
  (void)hipSetDevice(w->device);
  (void)hipStreamSynchronize(ls.stream);
  ls.harvest_timers();
  memset(out, 0, sizeof *out);
  out->cg_iters = ls.tot_cg_its;
  out->lin_sys_solves = ls.n_solves;
  out->mat_vecs = ls.n_matvecs;
  out->spmv_launches = ls.spmv_timer.samples; // launches that were event-timed
  out->spmv_ms = ls.spmv_timer.total_ms;
  out->cg_ms = ls.cg_timer.total_ms;
  out->nnz = ls.A.nnz;
  out->spmv_bytes = ls.matvec_bytes();
}

} // extern "C"
