// linsys.h -- device-resident Jacobi-PCG on the reduced KKT system
//     (R_x + P + A' R_y^{-1} A) x = r_x + A' R_y^{-1} r_y ,  y = R_y^{-1}(A x - r_y)
// i.e. the algorithm of reference linsys/cpu/indirect/private.c:133-324, with
// every vector in HBM and the loop controlled from the device.
#pragma once
#include "spmv.h"
#include "scs_host.h"
#include "spmv_wave.h"

namespace scsamd {

int selected_device(); // device chosen by scs_amd_set_device (HIP's current device is per host thread)

// Control block of one PCG solve, lives in device memory; the host reads it back
// once per enqueued batch of iterations.
struct CgCtl {
  real ztr[2];   // z'r, double-buffered by iteration parity
  real norm_r;   // ||r||_inf after the last completed iteration
  real tol;      // tolerance of this solve
  real rhs_norm; // ||b||_inf over n+m on entry
  int zero_rhs;  // ||b||_inf <= 1e-12: solution is 0, everything else is skipped
  int cg_done;   // converged (or breakdown): remaining iteration kernels return
  int iters;     // PCG iterations performed (reference counting, private.c:203,216)
  int max_its;   // 10 n (private.c:307): the device stops there even if more iterations were enqueued
};

// ONE linear system split by rows of A over several devices (the operator split is
// linsys/cpu/indirect/private.c:106-119).  A workspace created on the row slab A_r with diag_r = [R_x / N ; R_y of the slab]
// applies ITS term of G = R_x + A' R_y^-1 A = sum_r (R_x / N + A_r' R_r^-1 A_r); `allreduce` sums (or takes the maximum of)
// an n-vector over the N ranks IN PLACE, enqueued on the workspace's stream -- the PCG loop stays device controlled, the host
// reads one control block per batch of iterations as in the unsplit solve.  Everything of length n is replicated on every rank
// and computed redundantly (identical bits after the all-reduce), so alpha / beta / the stop test need no second collective.
struct ShardHook {
  void *ctx = nullptr;
  int world = 1, rank = 0;
  int (*allreduce)(void *ctx, real *buf, size_t count, int op /* 0 sum, 1 max */, hipStream_t st) = nullptr; // 0 = ok
};

struct LinSys {
  int n = 0, m = 0;
  ShardHook *shard = nullptr;   // non-null: this workspace holds one row slab (set_shard)
  long long n_allreduce = 0;
  EventTimer ar_timer;          // sampled all-reduce times (profiling)
  void set_shard(ShardHook *h); // call right after init(), before the first set_diag_r_*
  void shard_allreduce(real *buf, size_t count, int op);
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool has_P = false;
  bool use_fused = false; // whole solve in one workgroup (small systems)
  int nt_mode = 0;        // non-temporal policy of the update kernel's streams (SCS_AMD_VEC_NT; 0 = default policy; bit 2: two chunks per lane in flight)
  int dir_mode = 0;       // k_cg_direction: bit 0 non-temporal z reads, bit 1 non-temporal p stores, bit 2 two chunks per lane (SCS_AMD_DIR_MODE)
  bool use_cg2 = false;   // two launches per CG iteration (n <= CG2_N_MAX): k_cg2_a + transposed product
  bool use_cg3 = false;   // three launches per CG iteration (round 5): the stop test, alpha, beta, x, r, z, p in ONE vector kernel (k_cg3_update)

  CsrDev At; // CSR(A') == CSC(A): n rows, gathers an m-vector
  CsrDev A;  // CSR(A): m rows, gathers an n-vector
  CsrDev P;  // full symmetric CSR of P (n x n), optional
  DevBuf<real> Pdiag; // sum of stored diagonal entries of P per column

  DevBuf<real> rx, ry;               // R_x (n), R_y (m)
  DevBuf<real> M, p, r, Gp, z, Pp;   // n each
  DevBuf<real> p2, r2;               // second direction / residual buffers of the two-launch path (p_j, r_j in {p, p2}[j & 1])
  DevBuf<real> tmp;                  // m
  DevBuf<real> partA, partB;         // reduction partials
  DevBuf<real> partC, partD;         // use_cg3: partials of z'Gp and Gp'MGp
  DevBuf<CgCtl> ctl;
  PinnedBuf<CgCtl> hctl;

  // staging for the host-pointer boundary (B1)
  DevBuf<real> b_stage, s_stage, dr_stage;

  // HIP graph of CG_GRAPH_ITERS iterations (4 kernels each, constant arguments: everything that
  // changes lives in the control block) for systems whose kernels are shorter than a launch
  hipGraphExec_t cg_graph = nullptr;
  real *cg_x = nullptr; // solution vector of the current / captured solve
  bool cg_graph_tried = false, use_graph = false;
  void enqueue_cg_iteration(int q);
  void enqueue_cg2_iteration(long long it);
  void enqueue_cg3_iteration(long long it);
  bool build_cg_graph();
  // statistics / profiling
  long long tot_cg_its = 0, n_solves = 0, n_matvecs = 0, n_spmv = 0, n_graph_launches = 0;
  int last_its = 16;
  bool profiling = false;
  EventTimer spmv_timer, cg_timer;
  long long spmv_sample_ctr = 0;

  LinSys() = default;
  ~LinSys();
  LinSys(const LinSys &) = delete;

  // A, P: host CSC as handed over by the reference (normalized data).
  // `pat` (optional): an already-built pattern transpose of A_csc
  // with pat->dev.valid the matrices are adopted from HBM (device equilibration) instead of uploaded (round 5)
  void init(const CscView *A_csc, const CscView *P_csc, hipStream_t s, CsrPattern *pat = nullptr);
  // diag_r = [R_x (n); R_y (m)] : host or device source
  void set_diag_r_host(const real *diag_r);
  void set_diag_r_dev(const real *diag_r_dev);
  // Solve in place on device memory: b = [r_x; r_y] -> [x; y].  s: warm start
  // (device, length n) or nullptr.  If warm_part != nullptr the tolerance is
  // formed on the device as
  //     tol = max(1e-12, 0.2 * min(tol, max(warm_part[0..warm_cnt)) * warm_scale))
  // (reference src/scs.c:745-762), otherwise `tol` is used as given.
  // Returns the PCG iteration count (>= 0).
  int solve_dev(real *b, const real *s, real tol, const real *warm_part = nullptr,
                int warm_cnt = 0, real warm_scale = 0);
  // y_out(n) = (R_x + P + A' R_y^-1 A) x  -- exposed for tests / roofline runs
  void mat_vec_dev(const real *x, real *y_out, real *dot_partials);
  // plain products for the residual computation (reference src/scs.c:559,581)
  void mul_A(const real *x_n, real *y_m);  // y = A x
  void mul_At(const real *y_m, real *x_n); // x = A' y
  void mul_P(const real *x_n, real *y_n);  // y = P x (full symmetric)
  long long matvec_bytes() const { return A.algorithmic_bytes() + At.algorithmic_bytes(); }
  void harvest_timers();

private:
  void build_preconditioner();
  void launch_spmv(int epi, const CsrDev &mat, const real *x, real *y, const EpiArgs &e,
                   const int *skip);
};

} // namespace scsamd
