// admm.hip -- B2: scs_init / scs_update / scs_solve / scs_finish with the whole
// ADMM iteration resident in HBM.
//
// Control flow, constants and step order are those of reference src/scs.c:
//   scs_solve loop            :1356-1455     (order of AA, normalize_v, lin-sys,
//                                             cones, rsk, convergence, scale, dual)
//   project_lin_sys           :733-771       u_t, warm start, CG tolerance schedule
//   root_plus(_from_coeffs)   :689-730
//   project_cones             :796-810
//   compute_rsk / update_dual_vars / normalize_v   :781-821
//   populate_residual_struct / compute_residuals / unnormalize_residuals :463-607
//   has_converged             :611-649
//   update_scale / set_diag_r / update_work_cache  :1164-1241, :971-980, :1118-1128
//   warm/cold start, finalize, set_* status        :660-687, :825-969
// What differs is where things live: every length-(n+m+1) vector stays on the
// device for the whole solve; per iteration the host only launches kernels and
// reads back (i) the PCG control block, (ii) every 25 iterations ~20 reduced
// scalars for the convergence test.  Each glue step is a fused grid-stride
// kernel; reductions use the deterministic two-level scheme of common.h.
#include "linsys.h"
#include "cones.h"
#include "scs_host.h"
#include "reorder.h"
#include <csignal>
#include <mutex>
#include <chrono>
#include <algorithm>

namespace scsamd {

// ---- constants of include/glbopts.h that the loop uses ---------------------
static const int FEASIBLE_ITERS = 1;        // glbopts.h:188
static const int RESCALING_MIN_ITERS = 100; // glbopts.h:192
static const int CONVERGED_INTERVAL = 25;   // glbopts.h:206
static const int PRINT_INTERVAL = 250;      // glbopts.h:204
static const double TAU_FACTOR = 10.;       // glbopts.h:216
static const double MAX_SCALE_VALUE = 1e6, MIN_SCALE_VALUE = 1e-6; // glbopts.h:242-243
static const double CG_BEST_TOL = 1e-12;    // glbopts.h:247
static const double CG_RATE = 1.5;          // glbopts.h:257
static const double INFEAS_NEGATIVITY_TOL = 1e-9; // glbopts.h:52
static const double DIV_EPS_TOL = 1e-18;    // glbopts.h:194

static inline real safediv_pos(real x, real y) { return y < (real)DIV_EPS_TOL ? x / (real)DIV_EPS_TOL : x / y; }

constexpr int GLUE_MAX_GRID = 2048;
constexpr int NQ = 24; // reduced scalar slots
static inline int glue_grid(long long len) {
  long long g = (len + SCSAMD_BLOCK - 1) / SCSAMD_BLOCK;
  return (int)std::max<long long>(1, std::min<long long>(g, GLUE_MAX_GRID));
}

// ----------------------------------------------------------------------------
// glue kernels
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_sumsq_partial(const real *__restrict__ v, int len, real *part) {
  __shared__ real red[4];
  real s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) s += v[i] * v[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// normalize_v (:813-821) + v_prev copy (:1375-1377) + u_t = [R_x v; -R_y v; v_tau]
// (:738-744) + warm = u_x + tau g_x with its |.|_inf partials (:751-758)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_prep_linsys(real *v, real *v_prev, real *u_t,
                                                             const real *__restrict__ u,
                                                             const real *__restrict__ g,
                                                             const real *__restrict__ R, real *warm, int n,
                                                             int l, const real *nrm_part, int nrm_cnt,
                                                             real *warm_part, int do_normalize) {
  __shared__ real red[4];
  real factor = 1;
  if (do_normalize) {
    const real nrm = sqrt(reduce_partials_sum(nrm_part, nrm_cnt, red));
    if (nrm != (real)0) factor = sqrt((real)l) * (real)1. / nrm;
  }
  const real tau = u[l - 1];
  real mx = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < l; i += gridDim.x * blockDim.x) {
    real vi = v[i];
    if (do_normalize) {
      vi *= factor;
      v[i] = vi;
    }
    if (v_prev) v_prev[i] = vi;
    real ut;
    if (i < n) {
      ut = vi * R[i];
      const real w = u[i] + tau * g[i];
      warm[i] = w;
      const real a = absval(w);
      mx = a > mx ? a : mx;
    } else if (i < l - 1) {
      ut = -vi * R[i];
    } else {
      ut = vi;
    }
    u_t[i] = ut;
  }
  mx = block_max(mx, red);
  if (threadIdx.x == 0) warm_part[blockIdx.x] = mx;
}

// the five R-weighted dots of root_plus (:710-730); p = u_t, mu = v
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_root_plus_partial(const real *__restrict__ p,
                                                                    const real *__restrict__ mu,
                                                                    const real *__restrict__ g,
                                                                    const real *__restrict__ R, int nm,
                                                                    real *part, int stride) {
  __shared__ real red[4];
  real gg = 0, mug = 0, pg = 0, pp = 0, pmu = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += gridDim.x * blockDim.x) {
    const real ri = R[i], gi = g[i], pi = p[i], mui = mu[i];
    gg += gi * gi * ri;
    mug += mui * gi * ri;
    pg += pi * gi * ri;
    pp += pi * pi * ri;
    pmu += pi * mui * ri;
  }
  gg = block_sum(gg, red);
  mug = block_sum(mug, red);
  pg = block_sum(pg, red);
  pp = block_sum(pp, red);
  pmu = block_sum(pmu, red);
  if (threadIdx.x == 0) {
    part[0 * stride + blockIdx.x] = gg;
    part[1 * stride + blockIdx.x] = mug;
    part[2 * stride + blockIdx.x] = pg;
    part[3 * stride + blockIdx.x] = pp;
    part[4 * stride + blockIdx.x] = pmu;
  }
}

__device__ __forceinline__ real root_plus_from_coeffs(real a, real b, real c) { // :689-708
  if (!isfinite(a) || !isfinite(b) || !isfinite(c) || a <= (real)0) return (real)NAN;
  const real rad = b * b - 4 * a * c;
  if (!isfinite(rad)) return (real)NAN;
  if (rad < (real)0) return -b / (2 * a);
  const real sq = sqrt(rad);
  if (b <= (real)0) return (-b + sq) / (2 * a);
  const real q = (real)-0.5 * (b + sq);
  return q != (real)0 ? c / q : (real)0;
}

// tau~ (:764-768), u_t -= tau~ g (:769), u = 2 u_t - v (:798-800), and the Moreau
// pre-scaling of the cone part: cw = -R_y (2 u_t - v)_y  (src/cones.c:1568-1579)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_post_linsys(real *u_t, real *u, const real *__restrict__ v,
                                                              const real *__restrict__ g,
                                                              const real *__restrict__ R, real *cw, int n,
                                                              int l, const real *part, int cnt, int stride,
                                                              int feasible_iter) {
  __shared__ real red[4];
  real tau_t;
  if (feasible_iter) {
    tau_t = 1;
  } else {
    const real gg = reduce_partials_sum(part + 0 * stride, cnt, red);
    const real mug = reduce_partials_sum(part + 1 * stride, cnt, red);
    const real pg = reduce_partials_sum(part + 2 * stride, cnt, red);
    const real pp = reduce_partials_sum(part + 3 * stride, cnt, red);
    const real pmu = reduce_partials_sum(part + 4 * stride, cnt, red);
    const real tau_scale = R[l - 1], eta = v[l - 1];
    tau_t = root_plus_from_coeffs(tau_scale + gg, mug - 2 * pg - eta * tau_scale, pp - pmu);
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < l; i += gridDim.x * blockDim.x) {
    if (i < l - 1) {
      const real ut = u_t[i] + g[i] * (-tau_t);
      u_t[i] = ut;
      const real uu = 2 * ut - v[i];
      u[i] = uu;
      if (i >= n) cw[i - n] = uu * (-R[i]);
    } else {
      u_t[i] = tau_t;
      const real uu = 2 * tau_t - v[i];
      u[i] = feasible_iter ? (real)1 : (uu > (real)0 ? uu : (real)0); // :804-808
    }
  }
}

// Moreau post (src/cones.c:1585-1593), rsk = R (v + u - 2 u_t) (:781-786) and,
// when `alpha` > 0, the dual update v += alpha (u - u_t) (:788-793) fused in.
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_post_cone(real *u, const real *__restrict__ u_t, real *v,
                                                            real *rsk, const real *__restrict__ R,
                                                            const real *__restrict__ cw, int n, int l,
                                                            real alpha) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < l; i += gridDim.x * blockDim.x) {
    real ui = u[i];
    const real ri = R[i];
    if (i >= n && i < l - 1) {
      ui = cw[i - n] / ri + ui;
      u[i] = ui;
    }
    const real vi = v[i], ut = u_t[i];
    rsk[i] = (vi + ui - 2 * ut) * ri;
    if (alpha > (real)0) v[i] = vi + alpha * (ui - ut);
  }
}

__global__ __launch_bounds__(SCSAMD_BLOCK) void k_dual_update(real *v, const real *__restrict__ u,
                                                              const real *__restrict__ u_t, int l, real alpha) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < l; i += gridDim.x * blockDim.x)
    v[i] += alpha * (u[i] - u_t[i]);
}

// v <- rsk / R+ + 2 u_t - u  after a scale update (:1236-1238)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_remap_v(real *v, const real *__restrict__ rsk,
                                                          const real *__restrict__ R,
                                                          const real *__restrict__ u_t,
                                                          const real *__restrict__ u, int l) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < l; i += gridDim.x * blockDim.x)
    v[i] = rsk[i] / R[i] + 2 * u_t[i] - u[i];
}

// diag_r = [rho_x ... ; 1/(1000 scale) on the zero cone, 1/scale elsewhere ; TAU_FACTOR]
// (:971-980 + src/cones.c:349-363)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_set_diag_r(real *R, int n, int m, int z, real rho_x,
                                                             real scale) {
  const int l = n + m + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < l; i += gridDim.x * blockDim.x) {
    real r;
    if (i < n) r = rho_x;
    else if (i < n + z) r = (real)1.0 / ((real)1000. * scale);
    else if (i < l - 1) r = (real)1.0 / scale;
    else r = (real)TAU_FACTOR;
    R[i] = r;
  }
}

// g = [c; -b]   (:1118-1127)
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_build_g(real *g, const real *__restrict__ c,
                                                          const real *__restrict__ b, int n, int m) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + m; i += gridDim.x * blockDim.x)
    g[i] = i < n ? c[i] : -b[i - n];
}

// ---- residuals (:535-607 + :487-531): every vector norm / dot the host needs --
enum {
  Q_PRI_N = 0, Q_AXS_N, Q_AX_N, Q_PRI_O, Q_AXS_O, Q_AX_O, Q_S_O, Q_BTY, Q_S_N, // primal side
  Q_DUAL_N, Q_PX_N, Q_ATY_N, Q_DUAL_O, Q_PX_O, Q_ATY_O, Q_CTX, Q_XPX,           // dual side
  Q_TAU, Q_KAP, Q_COUNT
};
static_assert(Q_COUNT <= NQ, "slots");

__global__ __launch_bounds__(SCSAMD_BLOCK) void k_resid_primal(const real *__restrict__ ax,
                                                               const real *__restrict__ s,
                                                               const real *__restrict__ y,
                                                               const real *__restrict__ b,
                                                               const real *__restrict__ D, const real *tau_p,
                                                               real inv_ds, real ds, int m, real *part,
                                                               int stride) {
  __shared__ real red[4];
  const real tau = absval(*tau_p);
  real q_pri = 0, q_axs = 0, q_ax = 0, o_pri = 0, o_axs = 0, o_ax = 0, o_s = 0, n_s = 0, bty = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    const real axi = ax[i], si = s[i], bi = b[i], Di = D[i];
    const real axs = axi + si;
    const real pri = axs - tau * bi;
    const real f = inv_ds / Di;
    real a;
    a = absval(pri); q_pri = a > q_pri ? a : q_pri;
    a = absval(axs); q_axs = a > q_axs ? a : q_axs;
    a = absval(axi); q_ax = a > q_ax ? a : q_ax;
    a = absval(pri * f); o_pri = a > o_pri ? a : o_pri;
    a = absval(axs * f); o_axs = a > o_axs ? a : o_axs;
    a = absval(axi * f); o_ax = a > o_ax ? a : o_ax;
    a = absval(si / (Di * ds)); o_s = a > o_s ? a : o_s;
    a = absval(si); n_s = a > n_s ? a : n_s;
    bty += y[i] * bi;
  }
  q_pri = block_max(q_pri, red); q_axs = block_max(q_axs, red); q_ax = block_max(q_ax, red);
  o_pri = block_max(o_pri, red); o_axs = block_max(o_axs, red); o_ax = block_max(o_ax, red);
  o_s = block_max(o_s, red); n_s = block_max(n_s, red);
  bty = block_sum(bty, red);
  if (threadIdx.x == 0) {
    const int bx = blockIdx.x;
    part[Q_PRI_N * stride + bx] = q_pri; part[Q_AXS_N * stride + bx] = q_axs; part[Q_AX_N * stride + bx] = q_ax;
    part[Q_PRI_O * stride + bx] = o_pri; part[Q_AXS_O * stride + bx] = o_axs; part[Q_AX_O * stride + bx] = o_ax;
    part[Q_S_O * stride + bx] = o_s; part[Q_S_N * stride + bx] = n_s; part[Q_BTY * stride + bx] = bty;
  }
}

__global__ __launch_bounds__(SCSAMD_BLOCK) void k_resid_dual(const real *px, const real *__restrict__ aty,
                                                             const real *__restrict__ x,
                                                             const real *__restrict__ c,
                                                             const real *__restrict__ E, const real *tau_p,
                                                             real inv_ps, int n, real *part, int stride) {
  __shared__ real red[4];
  const real tau = absval(*tau_p);
  real q_d = 0, q_px = 0, q_aty = 0, o_d = 0, o_px = 0, o_aty = 0, ctx = 0, xpx = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const real pxi = px ? px[i] : (real)0, ai = aty[i], xi = x[i], ci = c[i];
    const real dual = pxi + ai + tau * ci;
    const real f = inv_ps / E[i];
    real a;
    a = absval(dual); q_d = a > q_d ? a : q_d;
    a = absval(pxi); q_px = a > q_px ? a : q_px;
    a = absval(ai); q_aty = a > q_aty ? a : q_aty;
    a = absval(dual * f); o_d = a > o_d ? a : o_d;
    a = absval(pxi * f); o_px = a > o_px ? a : o_px;
    a = absval(ai * f); o_aty = a > o_aty ? a : o_aty;
    ctx += xi * ci;
    xpx += pxi * xi;
  }
  q_d = block_max(q_d, red); q_px = block_max(q_px, red); q_aty = block_max(q_aty, red);
  o_d = block_max(o_d, red); o_px = block_max(o_px, red); o_aty = block_max(o_aty, red);
  ctx = block_sum(ctx, red); xpx = block_sum(xpx, red);
  if (threadIdx.x == 0) {
    const int bx = blockIdx.x;
    part[Q_DUAL_N * stride + bx] = q_d; part[Q_PX_N * stride + bx] = q_px; part[Q_ATY_N * stride + bx] = q_aty;
    part[Q_DUAL_O * stride + bx] = o_d; part[Q_PX_O * stride + bx] = o_px; part[Q_ATY_O * stride + bx] = o_aty;
    part[Q_CTX * stride + bx] = ctx; part[Q_XPX * stride + bx] = xpx;
  }
}

__global__ __launch_bounds__(SCSAMD_BLOCK) void k_resid_final(const real *part, int stride, int cnt_m,
                                                              int cnt_n, const real *tau_p,
                                                              const real *kap_p, real *out) {
  __shared__ real red[4];
  for (int q = 0; q < Q_TAU; ++q) {
    const bool primal = q < Q_DUAL_N;
    const int cnt = primal ? cnt_m : cnt_n;
    const bool is_sum = q == Q_BTY || q == Q_CTX || q == Q_XPX;
    const real r = is_sum ? reduce_partials_sum(part + q * stride, cnt, red)
                          : reduce_partials_max(part + q * stride, cnt, red);
    if (threadIdx.x == 0) out[q] = r;
  }
  if (threadIdx.x == 0) {
    out[Q_TAU] = absval(*tau_p);
    out[Q_KAP] = absval(*kap_p);
  }
}

} // namespace scsamd

// ============================================================================
// workspace
// ============================================================================
using namespace scsamd;

struct Resid { // reference include/scs_work.h:32-52 (scalars) + the norms the loop needs
  int last_iter = -1;
  real xt_p_x = 0, xt_p_x_tau = 0, ctx = 0, ctx_tau = 0, bty = 0, bty_tau = 0;
  real pobj = 0, dobj = 0, gap = 0, tau = 0, kap = 0;
  real res_pri = 0, res_dual = 0, res_infeas = 0, res_unbdd_p = 0, res_unbdd_a = 0;
  real nm_ax_s_btau = 0, nm_px_aty_ctau = 0, nm_ax_s = 0, nm_ax = 0, nm_px = 0, nm_aty = 0, nm_s = 0;
};

struct SCS_WORK {
  int n = 0, m = 0, l = 0, device = 0;
  ScsSettings stgs;
  // deep copies (host)
  std::vector<scs_int> cq, cs, ccs;
  std::vector<real> cbu, cbl, cpw;
  ScsCone k;
  HostCsc A, P;
  bool has_P = false;
  std::vector<real> b_orig, c_orig, b_nrm, c_nrm;
  real nm_b_orig = 0, nm_c_orig = 0;
  Scaling scal;
  double setup_time = 0;
  // device
  hipStream_t stream = nullptr;
  LinSys ls;
  ConeDev cone;
  DevBuf<real> u, u_t, v, v_prev, rsk, g, diag_r, b, c, D, E, warm, cw, ax, aty, px;
  DevBuf<real> part, qout;
  PinnedBuf<real> hq;
  long long psd_unconverged = 0;
  // residual / scale state
  Resid r_n, r_o;
  real sum_log_scale_factor = 0;
  int last_scale_update_iter = 0, n_log_scale_factor = 0, scale_updates = 0;
  int time_limit_reached = 0;
  // acceleration: device-resident for large l, host for small l (see init)
  AaHost *accel = nullptr;
  AaDev *accel_dev = nullptr;
  std::vector<real> hv, hv_prev;
  real aa_norm = 0;
  int rejected_accel_steps = 0, accepted_accel_steps = 0;
  // solve state (scs_solve = solve_begin + solve_steps + solve_end)
  int cur_iter = 0, run_status = SCS_UNFINISHED;
  bool loop_done = false, stepped = false;
  double t_solve0 = 0, t_lin = 0, t_accel = 0, cg_tol_override = 0;
  Reorder reord;                 // internal renumbering of variables / zero- and nonnegative-cone rows (reorder.h); mapped at the API boundary
  std::vector<real> hx_tmp, hy_tmp, hs_tmp; // staging for that mapping
  bool resid_every_iter = false; // scs_amd_set_residuals_every_iter: the cadence of a logged reference run (src/scs.c:1449-1454)
  // per-iteration CSV log (src/rw.c:686-863): diagnostic, computed on the host
  std::string log_csv_name;
  FILE *log_csv_fout = nullptr;
  std::vector<real> lg; // staging for the vectors the log takes norms of
  // instrumentation
  EventTimer cone_timer;
  bool profiling = false;
  long long cone_projs = 0;
  ~SCS_WORK() {
    if (log_csv_fout) fclose(log_csv_fout);
    if (accel) aa_host_finish(accel);
    if (accel_dev) aa_dev_finish(accel_dev);
    if (stream) {
      (void)hipStreamSynchronize(stream);
      (void)hipStreamDestroy(stream); // device buffers are freed by their own destructors
    }
  }
};

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- validation (src/scs.c:376-449) -------------------------------------------
static int validate_settings(const ScsSettings *s) {
#define BAD(cond, msg)                                                                             \
  if (cond) {                                                                                      \
    printf("%s\n", msg);                                                                           \
    return -1;                                                                                     \
  }
  BAD(s->max_iters <= 0, "max_iters must be positive");
  BAD(!std::isfinite((double)s->eps_abs) || s->eps_abs < 0, "eps_abs tolerance must be a nonnegative finite number");
  BAD(!std::isfinite((double)s->eps_rel) || s->eps_rel < 0, "eps_rel tolerance must be a nonnegative finite number");
  BAD(!std::isfinite((double)s->eps_infeas) || s->eps_infeas < 0, "eps_infeas tolerance must be a nonnegative finite number");
  BAD(!std::isfinite((double)s->alpha) || s->alpha <= 0 || s->alpha >= 2, "alpha must be in (0,2)");
  BAD(!std::isfinite((double)s->rho_x) || s->rho_x <= 0, "rho_x must be a positive finite number (1e-3 works well).");
  BAD(!std::isfinite((double)s->scale) || s->scale <= 0, "scale must be a positive finite number (1 works well).");
  BAD(!std::isfinite((double)s->time_limit_secs) || s->time_limit_secs < 0, "time_limit_secs must be a nonnegative finite number.");
  BAD(s->acceleration_interval <= 0, "acceleration_interval must be positive (10 works well).");
  BAD(s->acceleration_lookback < 0, "acceleration_lookback must be nonnegative (use acceleration_type_1=0 for type-II AA).");
  BAD(!std::isfinite((double)s->acceleration_regularization) || s->acceleration_regularization < 0,
      "acceleration_regularization must be a nonnegative finite number.");
  BAD(!std::isfinite((double)s->acceleration_relaxation) || s->acceleration_relaxation < 0 || s->acceleration_relaxation > 2,
      "acceleration_relaxation must be in [0, 2].");
#undef BAD
  return 0;
}

static int validate_problem(const ScsData *d, const ScsCone *k, const ScsSettings *s) {
  if (d->m <= 0 || d->n <= 0) {
    printf("m and n must both be greater than 0; m = %li, n = %li\n", (long)d->m, (long)d->n);
    return -1;
  }
  if (!d->A) {
    printf("A matrix missing\n");
    return -1;
  }
  if (!d->b || !d->c) {
    printf("b or c missing\n");
    return -1;
  }
  if (!fits_int32(d->A) || !fits_int32(d->P) || (long long)d->m + (long long)d->n + 1 > 2147483647LL) {
    // only a -DDLONG caller can get here: the device indexes with 32 bits
    printf(sizeof(eoff) == 8 ? "problem too large for the device indexing (m + n + 1 must stay below 2^31)\n"
                             : "problem too large for 32-bit device indexing (m + n + 1 and nnz must stay below 2^31; the DLONG build carries 64-bit entry positions)\n");
    return -1;
  }
  if (validate_csc(d->A, d->m, d->n, false, "A") < 0) return -1;
  if (d->P && validate_csc(d->P, d->n, d->n, true, "P") < 0) return -1;
  if (validate_cone(k, d->m, true) < 0) {
    printf("cone validation error\n");
    return -1;
  }
  return validate_settings(s);
}

// ---- device helpers --------------------------------------------------------------
static void set_diag_r(ScsWork *w) {
  hipLaunchKernelGGL(k_set_diag_r, dim3(glue_grid(w->l)), dim3(SCSAMD_BLOCK), 0, w->stream, w->diag_r.p, w->n,
                     w->m, w->k.z, w->stgs.rho_x, w->stgs.scale);
}

// g = (R + M)^-1 [c; -b]  to CG_BEST_TOL  (src/scs.c:1118-1128)
static void update_work_cache(ScsWork *w) {
  hipLaunchKernelGGL(k_build_g, dim3(glue_grid(w->n + w->m)), dim3(SCSAMD_BLOCK), 0, w->stream, w->g.p, w->c.p,
                     w->b.p, w->n, w->m);
  w->ls.solve_dev(w->g.p, nullptr, (real)CG_BEST_TOL);
}

static const int PSTRIDE = GLUE_MAX_GRID;

// populate_residual_struct (:535-607) -- vectors on the device, scalars to the host
static void compute_residuals_scalars(Resid *r, real pd) { // :463-485
  const real tol = (real)INFEAS_NEGATIVITY_TOL / pd;
  r->res_pri = safediv_pos(r->nm_ax_s_btau, r->tau);
  r->res_dual = safediv_pos(r->nm_px_aty_ctau, r->tau);
  r->res_unbdd_a = (real)NAN;
  r->res_unbdd_p = (real)NAN;
  r->res_infeas = (real)NAN;
  if (r->ctx_tau < -tol) {
    r->res_unbdd_a = safediv_pos(r->nm_ax_s, -r->ctx_tau);
    r->res_unbdd_p = safediv_pos(r->nm_px, -r->ctx_tau);
  }
  if (r->bty_tau < -tol) r->res_infeas = safediv_pos(r->nm_aty, -r->bty_tau);
}

static void populate_residuals(ScsWork *w, int iter) {
  if (w->r_n.last_iter == iter) return;
  const int n = w->n, m = w->m, l = w->l;
  hipStream_t st = w->stream;
  const real *x = w->u.p, *y = w->u.p + n, *s = w->rsk.p + n;
  w->ls.mul_A(x, w->ax.p);
  w->ls.mul_At(y, w->aty.p);
  if (w->has_P) w->ls.mul_P(x, w->px.p);
  const int gm = glue_grid(m), gn = glue_grid(n);
  const real ds = w->scal.dual_scale, ps = w->scal.primal_scale;
  hipLaunchKernelGGL(k_resid_primal, dim3(gm), dim3(SCSAMD_BLOCK), 0, st, w->ax.p, s, y, w->b.p, w->D.p,
                     w->u.p + l - 1, (real)1.0 / ds, ds, m, w->part.p, PSTRIDE);
  hipLaunchKernelGGL(k_resid_dual, dim3(gn), dim3(SCSAMD_BLOCK), 0, st, w->has_P ? w->px.p : (const real *)nullptr,
                     w->aty.p, x, w->c.p, w->E.p, w->u.p + l - 1, (real)1.0 / ps, n, w->part.p, PSTRIDE);
  hipLaunchKernelGGL(k_resid_final, dim3(1), dim3(SCSAMD_BLOCK), 0, st, w->part.p, PSTRIDE, gm, gn,
                     w->u.p + l - 1, w->rsk.p + l - 1, w->qout.p);
  HIP_CHECK(hipMemcpyAsync(w->hq.p, w->qout.p, NQ * sizeof(real), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  w->cone_timer.harvest(); // stream is idle here
  if (w->cone.n_psd > 0) { // the reference reports LAPACK info > 0 and carries on (src/cones.c:1031-1032, src/scs.c:1389)
    const int bad = w->cone.take_status(st);
    if (bad > 0 && w->psd_unconverged == 0 && w->stgs.verbose)
      printf("WARNING: %d PSD block projection(s) hit the Jacobi sweep cap (eigenvalues may be inaccurate)\n", bad);
    w->psd_unconverged += bad;
  }
  const real *q = w->hq.p;
  Resid &r = w->r_n;
  r.last_iter = iter;
  r.tau = q[Q_TAU];
  r.kap = q[Q_KAP];
  r.nm_ax_s_btau = q[Q_PRI_N];
  r.nm_ax_s = q[Q_AXS_N];
  r.nm_ax = q[Q_AX_N];
  r.nm_s = q[Q_S_N];
  r.nm_px_aty_ctau = q[Q_DUAL_N];
  r.nm_px = q[Q_PX_N];
  r.nm_aty = q[Q_ATY_N];
  r.xt_p_x_tau = w->has_P ? q[Q_XPX] : (real)0;
  r.bty_tau = q[Q_BTY];
  r.ctx_tau = q[Q_CTX];
  r.bty = safediv_pos(r.bty_tau, r.tau);
  r.ctx = safediv_pos(r.ctx_tau, r.tau);
  r.xt_p_x = safediv_pos(r.xt_p_x_tau, r.tau * r.tau);
  r.gap = std::fabs(r.xt_p_x + r.ctx + r.bty);
  r.pobj = r.xt_p_x / (real)2. + r.ctx;
  r.dobj = -r.xt_p_x / (real)2. - r.bty;
  compute_residuals_scalars(&r, (real)1.0);
  // unnormalize_residuals (:487-531)
  Resid &o = w->r_o;
  if (w->stgs.normalize) {
    const real pd = ps * ds;
    o.last_iter = iter;
    o.tau = r.tau;
    o.kap = r.kap / pd;
    o.bty_tau = r.bty_tau / pd;
    o.ctx_tau = r.ctx_tau / pd;
    o.xt_p_x_tau = r.xt_p_x_tau / pd;
    o.xt_p_x = r.xt_p_x / pd;
    o.ctx = r.ctx / pd;
    o.bty = r.bty / pd;
    o.pobj = r.pobj / pd;
    o.dobj = r.dobj / pd;
    o.gap = r.gap / pd;
    o.nm_ax_s_btau = q[Q_PRI_O];
    o.nm_ax_s = q[Q_AXS_O];
    o.nm_ax = q[Q_AX_O];
    o.nm_s = q[Q_S_O];
    o.nm_px_aty_ctau = q[Q_DUAL_O];
    o.nm_px = q[Q_PX_O];
    o.nm_aty = q[Q_ATY_O];
    compute_residuals_scalars(&o, pd);
  } else {
    o = r;
  }
}

// ---- per-iteration CSV log: column set, order and formats of src/rw.c:707-863 ----------
// A diagnostic (the reference recomputes all residuals every iteration when it is on),
// so the ~30 extra vector norms are taken on the host from downloaded copies instead of
// growing the hot-path kernels.  The header carries the five spectral column names the
// reference emits in its default (LAPACK) build although the rows never fill them.
static real h_norm_inf(const real *v, size_t len) {
  real a = 0;
  for (size_t i = 0; i < len; ++i) a = std::max(a, (real)std::fabs(v[i]));
  return a;
}
static real h_norm_2(const real *v, size_t len) {
  real a = 0;
  for (size_t i = 0; i < len; ++i) a += v[i] * v[i];
  return std::sqrt(a);
}
static void log_csv_row(ScsWork *w, int iter) {
  FILE *fout = w->log_csv_fout;
  if (!fout) return;
  populate_residuals(w, iter);
  const size_t n = (size_t)w->n, m = (size_t)w->m, l = (size_t)w->l;
  hipStream_t st = w->stream;
  // staging layout: u | u_t | v | v_prev | rsk | ax | aty | px | 6 work vectors
  w->lg.resize(5 * l + m + 2 * n + 3 * m + 3 * n);
  real *u = w->lg.data(), *u_t = u + l, *v = u_t + l, *v_prev = v + l, *rsk = v_prev + l, *ax = rsk + l,
       *aty = ax + m, *px = aty + n, *wm0 = px + n, *wm1 = wm0 + m, *wm2 = wm1 + m, *wn0 = wm2 + m, *wn1 = wn0 + n,
       *wn2 = wn1 + n;
  w->u.download(u, l, st);
  w->u_t.download(u_t, l, st);
  w->v.download(v, l, st);
  w->v_prev.download(v_prev, l, st);
  w->rsk.download(rsk, l, st);
  w->ax.download(ax, m, st);
  w->aty.download(aty, n, st);
  if (w->has_P) w->px.download(px, n, st);
  HIP_CHECK(hipStreamSynchronize(st));
  if (!w->has_P) std::fill(px, px + n, (real)0);
  const Resid &r = w->r_o, &rn = w->r_n;
  const real *xn = u, *yn = u + n, *sn = rsk + n;
  const bool nrm = w->stgs.normalize != 0;
  const real ds = w->scal.dual_scale, ps = w->scal.primal_scale;
  fprintf(fout, "%li,", (long)iter);
  fprintf(fout, "%.16e,", (double)r.res_pri);
  fprintf(fout, "%.16e,", (double)r.res_dual);
  fprintf(fout, "%.16e,", (double)r.gap);
  { // un-normalised iterate (src/normalize.c:78-91)
    for (size_t j = 0; j < n; ++j) wn0[j] = nrm ? xn[j] * (w->scal.E[j] / ds) : xn[j];
    for (size_t i = 0; i < m; ++i) {
      wm0[i] = nrm ? yn[i] * (w->scal.D[i] / ps) : yn[i];
      wm1[i] = nrm ? sn[i] / (w->scal.D[i] * ds) : sn[i];
    }
    fprintf(fout, "%.16e,", (double)h_norm_inf(wn0, n));
    fprintf(fout, "%.16e,", (double)h_norm_inf(wm0, m));
    fprintf(fout, "%.16e,", (double)h_norm_inf(wm1, m));
    fprintf(fout, "%.16e,", (double)h_norm_2(wn0, n));
    fprintf(fout, "%.16e,", (double)h_norm_2(wm0, m));
    fprintf(fout, "%.16e,", (double)h_norm_2(wm1, m));
  }
  fprintf(fout, "%.16e,", (double)h_norm_inf(xn, n));
  fprintf(fout, "%.16e,", (double)h_norm_inf(yn, m));
  fprintf(fout, "%.16e,", (double)h_norm_inf(sn, m));
  fprintf(fout, "%.16e,", (double)h_norm_2(xn, n));
  fprintf(fout, "%.16e,", (double)h_norm_2(yn, m));
  fprintf(fout, "%.16e,", (double)h_norm_2(sn, m));
  // residual vectors, normalised (wm2 / wn2) and original (wm0.. / wn0..)
  const real inv_ds = (real)1.0 / ds, inv_ps = (real)1.0 / ps;
  for (size_t i = 0; i < m; ++i) {
    const real axs = ax[i] + sn[i];
    wm2[i] = axs - rn.tau * w->b_nrm[i];
    const real f = nrm ? inv_ds / w->scal.D[i] : (real)1;
    wm0[i] = wm2[i] * f; // ax_s_btau
    wm1[i] = axs * f;    // ax_s
  }
  for (size_t j = 0; j < n; ++j) {
    wn2[j] = px[j] + aty[j] + rn.tau * w->c_nrm[j];
    const real f = nrm ? inv_ps / w->scal.E[j] : (real)1;
    wn0[j] = wn2[j] * f; // px_aty_ctau
  }
  fprintf(fout, "%.16e,", (double)h_norm_inf(wm0, m));
  fprintf(fout, "%.16e,", (double)h_norm_inf(wn0, n));
  fprintf(fout, "%.16e,", (double)h_norm_2(wm0, m));
  fprintf(fout, "%.16e,", (double)h_norm_2(wn0, n));
  fprintf(fout, "%.16e,", (double)r.res_infeas);
  fprintf(fout, "%.16e,", (double)r.res_unbdd_a);
  fprintf(fout, "%.16e,", (double)r.res_unbdd_p);
  fprintf(fout, "%.16e,", (double)r.pobj);
  fprintf(fout, "%.16e,", (double)r.dobj);
  fprintf(fout, "%.16e,", (double)r.tau);
  fprintf(fout, "%.16e,", (double)r.kap);
  fprintf(fout, "%.16e,", (double)rn.res_pri);
  fprintf(fout, "%.16e,", (double)rn.res_dual);
  fprintf(fout, "%.16e,", (double)rn.gap);
  fprintf(fout, "%.16e,", (double)h_norm_inf(wm2, m));
  fprintf(fout, "%.16e,", (double)h_norm_inf(wn2, n));
  fprintf(fout, "%.16e,", (double)h_norm_2(wm2, m));
  fprintf(fout, "%.16e,", (double)h_norm_2(wn2, n));
  fprintf(fout, "%.16e,", (double)rn.res_infeas);
  fprintf(fout, "%.16e,", (double)rn.res_unbdd_a);
  fprintf(fout, "%.16e,", (double)rn.res_unbdd_p);
  fprintf(fout, "%.16e,", (double)rn.pobj);
  fprintf(fout, "%.16e,", (double)rn.dobj);
  fprintf(fout, "%.16e,", (double)rn.tau);
  fprintf(fout, "%.16e,", (double)rn.kap);
  { // ax, ax_s, px, aty of the original problem
    for (size_t i = 0; i < m; ++i) wm2[i] = ax[i] * (nrm ? inv_ds / w->scal.D[i] : (real)1);
    fprintf(fout, "%.16e,", (double)h_norm_inf(wm2, m));
    fprintf(fout, "%.16e,", (double)h_norm_inf(wm1, m));
    for (size_t j = 0; j < n; ++j) {
      const real f = nrm ? inv_ps / w->scal.E[j] : (real)1;
      wn1[j] = px[j] * f;
      wn2[j] = aty[j] * f;
    }
    fprintf(fout, "%.16e,", (double)h_norm_inf(wn1, n));
    fprintf(fout, "%.16e,", (double)h_norm_inf(wn2, n));
  }
  fprintf(fout, "%.16e,", (double)r.xt_p_x);
  fprintf(fout, "%.16e,", (double)r.xt_p_x_tau);
  fprintf(fout, "%.16e,", (double)r.ctx);
  fprintf(fout, "%.16e,", (double)r.ctx_tau);
  fprintf(fout, "%.16e,", (double)r.bty);
  fprintf(fout, "%.16e,", (double)r.bty_tau);
  fprintf(fout, "%.16e,", (double)h_norm_inf(w->b_orig.data(), m));
  fprintf(fout, "%.16e,", (double)h_norm_inf(w->c_orig.data(), n));
  fprintf(fout, "%.16e,", (double)w->stgs.scale);
  {
    real d2u = 0, d2v = 0, diu = 0, div = 0;
    for (size_t i = 0; i < l; ++i) {
      const real a = u[i] - u_t[i], b = v[i] - v_prev[i];
      d2u += a * a;
      d2v += b * b;
      diu = std::max(diu, (real)std::fabs(a));
      div = std::max(div, (real)std::fabs(b));
    }
    fprintf(fout, "%.16e,", (double)std::sqrt(d2u));
    fprintf(fout, "%.16e,", (double)std::sqrt(d2v));
    fprintf(fout, "%.16e,", (double)diu);
    fprintf(fout, "%.16e,", (double)div);
  }
  fprintf(fout, "%.16e,", (double)w->aa_norm);
  fprintf(fout, "%li,", (long)w->accepted_accel_steps);
  fprintf(fout, "%li,", (long)w->rejected_accel_steps);
  fprintf(fout, "%.16e,", (now_ms() - w->t_solve0) / 1e3);
  fprintf(fout, "\n");
}
static const char *LOG_CSV_HEADER =
    "iter,res_pri,res_dual,gap,x_nrm_inf,y_nrm_inf,s_nrm_inf,x_nrm_2,y_nrm_2,s_nrm_2,x_nrm_inf_normalized,"
    "y_nrm_inf_normalized,s_nrm_inf_normalized,x_nrm_2_normalized,y_nrm_2_normalized,s_nrm_2_normalized,"
    "ax_s_btau_nrm_inf,px_aty_ctau_nrm_inf,ax_s_btau_nrm_2,px_aty_ctau_nrm_2,res_infeas,res_unbdd_a,res_unbdd_p,"
    "pobj,dobj,tau,kap,res_pri_normalized,res_dual_normalized,gap_normalized,ax_s_btau_nrm_inf_normalized,"
    "px_aty_ctau_nrm_inf_normalized,ax_s_btau_nrm_2_normalized,px_aty_ctau_nrm_2_normalized,res_infeas_normalized,"
    "res_unbdd_a_normalized,res_unbdd_p_normalized,pobj_normalized,dobj_normalized,tau_normalized,kap_normalized,"
    "ax_nrm_inf,ax_s_nrm_inf,px_nrm_inf,aty_nrm_inf,xt_p_x,xt_p_x_tau,ctx,ctx_tau,bty,bty_tau,b_nrm_inf,c_nrm_inf,"
    "scale,diff_u_ut_nrm_2,diff_v_v_prev_nrm_2,diff_u_ut_nrm_inf,diff_v_v_prev_nrm_inf,aa_norm,"
    "accepted_accel_steps,rejected_accel_steps,time,spectral_Newton_iter,plain_Newton_success,res_dual_spectral,"
    "res_pri_spectral,comp_spectral,\n";

static int has_converged(ScsWork *w) { // :611-649
  const Resid &r = w->r_o;
  const real eps_abs = w->stgs.eps_abs, eps_rel = w->stgs.eps_rel, eps_infeas = w->stgs.eps_infeas;
  if (r.tau > (real)0.) {
    const real grl = std::max(std::max(std::fabs(r.xt_p_x), std::fabs(r.ctx)), std::fabs(r.bty));
    const real prl = std::max(std::max(w->nm_b_orig * r.tau, r.nm_s), r.nm_ax) / r.tau;
    const real drl = std::max(std::max(w->nm_c_orig * r.tau, r.nm_px), r.nm_aty) / r.tau;
    if (std::isless(r.res_pri, eps_abs + eps_rel * prl) && std::isless(r.res_dual, eps_abs + eps_rel * drl) &&
        std::isless(r.gap, eps_abs + eps_rel * grl))
      return SCS_SOLVED;
  }
  if (std::isless(r.res_unbdd_a, eps_infeas) && std::isless(r.res_unbdd_p, eps_infeas)) return SCS_UNBOUNDED;
  if (std::isless(r.res_infeas, eps_infeas)) return SCS_INFEASIBLE;
  return 0;
}

static int update_scale(ScsWork *w, int iter) { // :1164-1241
  const Resid &r = w->r_o;
  const int since = iter - w->last_scale_update_iter;
  real denom_pri = std::max(r.nm_ax, r.nm_s);
  denom_pri = std::max(denom_pri, w->nm_b_orig * r.tau);
  real rel_pri = safediv_pos(r.nm_ax_s_btau, denom_pri);
  real denom_dual = std::max(r.nm_px, r.nm_aty);
  denom_dual = std::max(denom_dual, w->nm_c_orig * r.tau);
  real rel_dual = safediv_pos(r.nm_px_aty_ctau, denom_dual);
  rel_pri = std::max(rel_pri, (real)DIV_EPS_TOL);
  rel_dual = std::max(rel_dual, (real)DIV_EPS_TOL);
  w->sum_log_scale_factor += std::log(rel_pri) - std::log(rel_dual);
  w->n_log_scale_factor++;
  const real factor = std::sqrt(std::exp(w->sum_log_scale_factor / (real)w->n_log_scale_factor));
  if (since < RESCALING_MIN_ITERS) return 0;
  const real new_scale =
      std::min(std::max(w->stgs.scale * factor, (real)MIN_SCALE_VALUE), (real)MAX_SCALE_VALUE);
  if (new_scale == w->stgs.scale) return 0;
  if (factor > std::sqrt((real)10.) || factor < (real)1. / std::sqrt((real)10.)) { // :1160-1162
    w->scale_updates++;
    w->sum_log_scale_factor = 0;
    w->n_log_scale_factor = 0;
    w->last_scale_update_iter = iter;
    w->stgs.scale = new_scale;
    set_diag_r(w);
    w->ls.set_diag_r_dev(w->diag_r.p);
    update_work_cache(w);
    if (w->accel) aa_host_reset(w->accel);
    if (w->accel_dev) aa_dev_reset(w->accel_dev);
    hipLaunchKernelGGL(k_remap_v, dim3(glue_grid(w->l)), dim3(SCSAMD_BLOCK), 0, w->stream, w->v.p, w->rsk.p,
                       w->diag_r.p, w->u_t.p, w->u.p, w->l);
  }
  return 0;
}

// ---- printing: the reference's layout (src/scs.c:113-272), line for line; only the banner names this backend --
extern "C" char *_scs_get_cone_header(const ScsCone *k); // cones_shim.cpp (src/cones.c:565-581)
static void print_rule() {
  for (int i = 0; i < 66; ++i) putchar('-'); // LINE_LEN, scs.c:121
  putchar('\n');
}
static void print_header(const ScsWork *w) {
  print_rule();
  printf("\t       SCS v%s - Splitting Conic Solver\n\tMI355X (gfx950) ADMM hot path, device-resident (scs_amd)\n", scs_version());
  print_rule();
  printf("problem:  variables n: %i, constraints m: %i\n", (int)w->n, (int)w->m);
  if (char *cs = _scs_get_cone_header(&w->k)) {
    printf("%s", cs);
    free(cs);
  } else {
    printf("cones: <unavailable>\n");
  }
  printf("settings: eps_abs: %.1e, eps_rel: %.1e, eps_infeas: %.1e\n"
         "\t  alpha: %.2f, scale: %.2e, adaptive_scale: %i\n"
         "\t  max_iters: %i, normalize: %i, rho_x: %.2e\n",
         (double)w->stgs.eps_abs, (double)w->stgs.eps_rel, (double)w->stgs.eps_infeas, (double)w->stgs.alpha,
         (double)w->stgs.scale, (int)w->stgs.adaptive_scale, (int)w->stgs.max_iters, (int)w->stgs.normalize,
         (double)w->stgs.rho_x);
  if (w->stgs.acceleration_lookback != 0)
    printf("\t  acceleration_lookback: %i, acceleration_interval: %i\n", (int)w->stgs.acceleration_lookback,
           (int)w->stgs.acceleration_interval);
  if (w->stgs.time_limit_secs) printf("\t  time_limit_secs: %.2e\n", (double)w->stgs.time_limit_secs);
  printf("lin-sys:  %s\n\t  nnz(A): %li, nnz(P): %li\n", scs_get_lin_sys_method(), (long)w->A.p[w->n],
         w->has_P ? (long)w->P.p[w->n] : 0l);
  print_rule();
  printf(" iter | pri res | dua res |   gap   |   obj   |  scale  | time (s)\n");
  print_rule();
}
static void print_summary(const ScsWork *w, int i, double t0) {
  const Resid &r = w->r_o;
  // scs.c:207-219: total time including setup
  printf("%*i|%*.2e %*.2e %*.2e %*.2e %*.2e %*.2e \n", 6, i, 9, (double)r.res_pri, 9, (double)r.res_dual, 9, (double)r.gap, 9,
         (double)(0.5 * (r.pobj + r.dobj)), 9, (double)w->stgs.scale, 9, (now_ms() - t0 + w->setup_time) / 1e3);
  fflush(stdout);
}
static void print_footer(const ScsInfo *info) { // scs.c:246-272
  print_rule();
  printf("status:  %s\n", info->status);
  printf("timings: total: %1.2es = setup: %1.2es + solve: %1.2es\n", (double)(info->setup_time + info->solve_time) / 1e3,
         (double)info->setup_time / 1e3, (double)info->solve_time / 1e3);
  printf("\t lin-sys: %1.2es, cones: %1.2es, accel: %1.2es\n", (double)info->lin_sys_time / 1e3,
         (double)info->cone_time / 1e3, (double)info->accel_time / 1e3);
  print_rule();
  printf("objective = %.6f", (double)(0.5 * (info->pobj + info->dobj)));
  if (info->status_val == SCS_SOLVED_INACCURATE || info->status_val == SCS_UNBOUNDED_INACCURATE ||
      info->status_val == SCS_INFEASIBLE_INACCURATE)
    printf(" (inaccurate)");
  printf("\n");
  print_rule();
}

// ---- solution extraction (:825-969) ----------------------------------------------
static void fill_nan(real *p, int len) {
  for (int i = 0; i < len; ++i) p[i] = (real)NAN;
}
static void scale_vec(real *p, real a, int len) {
  for (int i = 0; i < len; ++i) p[i] *= a;
}
static void set_solved(const ScsWork *w, ScsSolution *sol, ScsInfo *info) {
  const real it = safediv_pos((real)1.0, w->r_o.tau);
  scale_vec(sol->x, it, w->n);
  scale_vec(sol->y, it, w->m);
  scale_vec(sol->s, it, w->m);
  info->gap = w->r_o.gap;
  info->res_pri = w->r_o.res_pri;
  info->res_dual = w->r_o.res_dual;
  info->pobj = w->r_o.xt_p_x / (real)2. + w->r_o.ctx;
  info->dobj = -w->r_o.xt_p_x / (real)2. - w->r_o.bty;
  strcpy(info->status, "solved");
  info->status_val = SCS_SOLVED;
}
static void set_infeasible(const ScsWork *w, ScsSolution *sol, ScsInfo *info) {
  scale_vec(sol->y, (real)-1 / w->r_o.bty_tau, w->m);
  fill_nan(sol->x, w->n);
  fill_nan(sol->s, w->m);
  info->gap = info->res_pri = info->res_dual = (real)NAN;
  info->pobj = info->dobj = (real)INFINITY;
  strcpy(info->status, "infeasible");
  info->status_val = SCS_INFEASIBLE;
}
static void set_unbounded(const ScsWork *w, ScsSolution *sol, ScsInfo *info) {
  scale_vec(sol->x, (real)-1 / w->r_o.ctx_tau, w->n);
  scale_vec(sol->s, (real)-1 / w->r_o.ctx_tau, w->m);
  fill_nan(sol->y, w->m);
  info->gap = info->res_pri = info->res_dual = (real)NAN;
  info->pobj = info->dobj = -(real)INFINITY;
  strcpy(info->status, "unbounded");
  info->status_val = SCS_UNBOUNDED;
}
static void set_unfinished(const ScsWork *w, ScsSolution *sol, ScsInfo *info) {
  const Resid &r = w->r_o;
  if (r.kap > r.tau && (r.bty_tau < 0 || r.ctx_tau < 0)) {
    if (r.bty_tau < 0 && r.bty_tau < r.ctx_tau) {
      set_infeasible(w, sol, info);
      info->status_val = SCS_INFEASIBLE_INACCURATE;
    } else {
      set_unbounded(w, sol, info);
      info->status_val = SCS_UNBOUNDED_INACCURATE;
    }
  } else if (r.tau > 0) {
    set_solved(w, sol, info);
    info->status_val = SCS_SOLVED_INACCURATE;
  } else {
    printf("ERROR: could not determine problem status.\n");
    info->status_val = SCS_FAILED;
  }
  if (w->time_limit_reached) strcat(info->status, " (inaccurate - reached time_limit_secs)");
  else if (info->iter >= w->stgs.max_iters || w->stepped) strcat(info->status, " (inaccurate - reached max_iters)");
  else printf("ERROR: should not be in this state (1).\n");
}

static void finalize(ScsWork *w, ScsSolution *sol, ScsInfo *info, int iter) {
  const int n = w->n, m = w->m;
  if (!sol->x) sol->x = (real *)calloc(n, sizeof(real));
  if (!sol->y) sol->y = (real *)calloc(m, sizeof(real));
  if (!sol->s) sol->s = (real *)calloc(m, sizeof(real));
  if (w->reord.active) { // our numbering -> the caller's, after the un-normalisation (D, E are in our numbering)
    w->hx_tmp.resize(n);
    w->hy_tmp.resize(m);
    w->hs_tmp.resize(m);
    HIP_CHECK(hipMemcpyAsync(w->hx_tmp.data(), w->u.p, n * sizeof(real), hipMemcpyDeviceToHost, w->stream));
    HIP_CHECK(hipMemcpyAsync(w->hy_tmp.data(), w->u.p + n, m * sizeof(real), hipMemcpyDeviceToHost, w->stream));
    HIP_CHECK(hipMemcpyAsync(w->hs_tmp.data(), w->rsk.p + n, m * sizeof(real), hipMemcpyDeviceToHost, w->stream));
    HIP_CHECK(hipStreamSynchronize(w->stream));
    if (w->stgs.normalize) un_normalize_sol(w->scal, w->hx_tmp.data(), w->hy_tmp.data(), w->hs_tmp.data());
    for (int j = 0; j < n; ++j) sol->x[w->reord.col_new2old[j]] = w->hx_tmp[j];
    for (int i = 0; i < m; ++i) {
      sol->y[w->reord.row_new2old[i]] = w->hy_tmp[i];
      sol->s[w->reord.row_new2old[i]] = w->hs_tmp[i];
    }
  } else {
    HIP_CHECK(hipMemcpyAsync(sol->x, w->u.p, n * sizeof(real), hipMemcpyDeviceToHost, w->stream));
    HIP_CHECK(hipMemcpyAsync(sol->y, w->u.p + n, m * sizeof(real), hipMemcpyDeviceToHost, w->stream));
    HIP_CHECK(hipMemcpyAsync(sol->s, w->rsk.p + n, m * sizeof(real), hipMemcpyDeviceToHost, w->stream));
    HIP_CHECK(hipStreamSynchronize(w->stream));
    if (w->stgs.normalize) un_normalize_sol(w->scal, sol->x, sol->y, sol->s);
  }
  populate_residuals(w, iter);
  real nm_s = 0, nm_y = 0, sty = 0;
  for (int i = 0; i < m; ++i) {
    nm_s = std::max(nm_s, (real)std::fabs(sol->s[i]));
    nm_y = std::max(nm_y, (real)std::fabs(sol->y[i]));
    sty += sol->s[i] * sol->y[i];
  }
  info->setup_time = (real)w->setup_time;
  info->iter = iter;
  info->res_infeas = w->r_o.res_infeas;
  info->res_unbdd_a = w->r_o.res_unbdd_a;
  info->res_unbdd_p = w->r_o.res_unbdd_p;
  info->scale = w->stgs.scale;
  info->scale_updates = w->scale_updates;
  info->rejected_accel_steps = w->rejected_accel_steps;
  info->accepted_accel_steps = w->accepted_accel_steps;
  memset(&info->aa_stats, 0, sizeof info->aa_stats);
  info->aa_stats.last_aa_norm = (real)NAN;
  if (w->accel) aa_host_stats(w->accel, &info->aa_stats);
  if (w->accel_dev) aa_dev_stats(w->accel_dev, &info->aa_stats);
  info->comp_slack = std::fabs(sty);
  if (info->comp_slack > (real)1e-5 * std::max(nm_s, nm_y))
    printf("WARNING - large complementary slackness residual: %f\n", (double)info->comp_slack);
  switch (info->status_val) {
  case SCS_SOLVED: set_solved(w, sol, info); break;
  case SCS_INFEASIBLE: set_infeasible(w, sol, info); break;
  case SCS_UNBOUNDED: set_unbounded(w, sol, info); break;
  case SCS_UNFINISHED: set_unfinished(w, sol, info); break;
  default: printf("ERROR: should not be in this state (2).\n");
  }
}

// ---- ctrl-c support (behaviour of src/ctrlc.c:84-125): while at least one solve is in
// flight SIGINT sets a flag instead of killing the process; the loop polls it at the
// convergence cadence and returns SCS_SIGINT.  Counted, so concurrent solves on several
// threads install / restore the previous handler exactly once.
namespace {
volatile sig_atomic_t g_sigint_seen = 0;
struct sigaction g_sigint_prev;
std::mutex g_sigint_mu;
int g_sigint_listeners = 0;
static void on_sigint(int sig) { g_sigint_seen = sig ? sig : -1; }
struct InterruptListener { // RAII: every exit path of scs_solve restores the handler
  InterruptListener() {
    std::lock_guard<std::mutex> lk(g_sigint_mu);
    if (g_sigint_listeners++ == 0) {
      g_sigint_seen = 0;
      struct sigaction act;
      memset(&act, 0, sizeof act);
      sigemptyset(&act.sa_mask);
      act.sa_handler = on_sigint;
      sigaction(SIGINT, &act, &g_sigint_prev);
    }
  }
  ~InterruptListener() {
    std::lock_guard<std::mutex> lk(g_sigint_mu);
    if (g_sigint_listeners > 0 && --g_sigint_listeners == 0) sigaction(SIGINT, &g_sigint_prev, nullptr);
  }
};
inline bool interrupted() { return g_sigint_seen != 0; }
} // namespace

static scs_int fail_out(ScsWork *w, int m, int n, ScsSolution *sol, ScsInfo *info, scs_int status,
                        const char *msg, const char *ststr) { // :321-371
  if (info) {
    info->gap = info->res_pri = info->res_dual = info->pobj = info->dobj = (real)NAN;
    info->iter = -1;
    info->status_val = status;
    info->solve_time = (real)NAN;
    strcpy(info->status, ststr);
  }
  if (sol) {
    if (n > 0) {
      if (!sol->x) sol->x = (real *)calloc(n, sizeof(real));
      fill_nan(sol->x, n);
    }
    if (m > 0) {
      if (!sol->y) sol->y = (real *)calloc(m, sizeof(real));
      fill_nan(sol->y, m);
      if (!sol->s) sol->s = (real *)calloc(m, sizeof(real));
      fill_nan(sol->s, m);
    }
  }
  printf("Failure:%s\n", msg);
  return status;
}

// ============================================================================
// public API
// ============================================================================
extern "C" {

const char *scs_version(void) { return "3.2.11-amd-gfx950-r1"; }

void scs_set_default_settings(ScsSettings *s) { // src/util.c:158-179, include/glbopts.h:35-50
  s->max_iters = 100000;
  s->eps_abs = (real)1e-4;
  s->eps_rel = (real)1e-4;
  s->eps_infeas = (real)1e-7;
  s->alpha = (real)1.5;
  s->rho_x = (real)1e-6;
  s->scale = (real)0.1;
  s->verbose = 1;
  s->normalize = 1;
  s->warm_start = 0;
  s->acceleration_lookback = 10;
  s->acceleration_interval = 10;
  s->acceleration_type_1 = 1;
  s->acceleration_regularization = (real)1e-8;
  s->acceleration_relaxation = (real)1.0;
  s->adaptive_scale = 1;
  s->write_data_filename = nullptr;
  s->log_csv_filename = nullptr;
  s->time_limit_secs = 0;
}

scs_int scs_update(ScsWork *w, scs_float *b, scs_float *c) { // src/scs.c:1287-1325
  if (!w) return -1;
  const double t0 = now_ms();
  try {
    HIP_CHECK(hipSetDevice(w->device));
    if (b) {
      if (w->b_orig.data() != b) { // (scs_init passes its own, already renumbered, copy)
        if (w->reord.active) for (int i = 0; i < w->m; ++i) w->b_orig[i] = b[w->reord.row_new2old[i]];
        else std::copy(b, b + w->m, w->b_orig.begin());
      }
      real nb = 0;
      for (int i = 0; i < w->m; ++i) nb = std::max(nb, (real)std::fabs(w->b_orig[i]));
      w->nm_b_orig = nb;
    }
    if (c) {
      if (w->c_orig.data() != c) {
        if (w->reord.active) for (int j = 0; j < w->n; ++j) w->c_orig[j] = c[w->reord.col_new2old[j]];
        else std::copy(c, c + w->n, w->c_orig.begin());
      }
      real nc = 0;
      for (int i = 0; i < w->n; ++i) nc = std::max(nc, (real)std::fabs(w->c_orig[i]));
      w->nm_c_orig = nc;
    }
    w->b_nrm = w->b_orig;
    w->c_nrm = w->c_orig;
    if (w->stgs.normalize) normalize_b_c(w->scal, w->b_nrm.data(), w->c_nrm.data());
    w->b.upload(w->b_nrm.data(), w->m, w->stream);
    w->c.upload(w->c_nrm.data(), w->n, w->stream);
    HIP_CHECK(hipStreamSynchronize(w->stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  w->setup_time = now_ms() - t0;
  return 0;
}

ScsWork *scs_init(const ScsData *d, const ScsCone *k, const ScsSettings *stgs) { // :1245-1285, :982-1116
  if (!d || !k || !stgs) {
    printf("ERROR: Missing ScsData, ScsCone, or ScsSettings input\n");
    return nullptr;
  }
  if (validate_problem(d, k, stgs) < 0) {
    printf("ERROR: Validation returned failure\n");
    return nullptr;
  }
  const double t0 = now_ms();
  ScsWork *w = nullptr;
  try {
    if (scs_amd_device_count() <= 0)
      throw HipError("scs_amd: no HIP device visible -- this backend has no CPU fallback");
    const int dev = selected_device(); // snapshot once: another thread may re-select concurrently
    HIP_CHECK(hipSetDevice(dev));
    w = new ScsWork();
    w->device = dev;
    const int n = w->n = d->n, m = w->m = d->m, l = w->l = d->n + d->m + 1;
    w->stgs = *stgs;
    if (stgs->write_data_filename) { // src/scs.c:1272-1275
      printf("Writing raw problem data to %s\n", stgs->write_data_filename);
      write_problem(d, k, stgs, stgs->write_data_filename);
    }
    if (stgs->log_csv_filename) { // src/scs.c:1275-1278
      printf("Logging run data to %s\n", stgs->log_csv_filename);
      w->log_csv_name = stgs->log_csv_filename;
    }
    w->stgs.write_data_filename = nullptr;
    w->stgs.log_csv_filename = nullptr;
    // deep copies
    w->k = *k;
    if (k->qsize) w->cq.assign(k->q, k->q + k->qsize);
    if (k->ssize) w->cs.assign(k->s, k->s + k->ssize);
    if (k->bsize > 1) {
      w->cbu.assign(k->bu, k->bu + k->bsize - 1);
      w->cbl.assign(k->bl, k->bl + k->bsize - 1);
    }
    w->k.q = w->cq.data();
    w->k.s = w->cs.data();
    w->k.bu = w->cbu.data();
    w->k.bl = w->cbl.data();
    if (k->cssize) w->ccs.assign(k->cs, k->cs + k->cssize);
    w->k.cs = w->ccs.data();
    if (k->psize) w->cpw.assign(k->p, k->p + k->psize);
    w->k.p = w->cpw.data();
    const bool dbg = opt_get("debug") != nullptr;
    double tp = now_ms();
    auto phase = [&](const char *what) {
      if (dbg) fprintf(stderr, "[scs_amd init] %-22s %8.1f ms\n", what, now_ms() - tp);
      tp = now_ms();
    };
    w->A.copy_from(d->A);
    w->has_P = d->P != nullptr;
    if (w->has_P) w->P.copy_from(d->P);
    // locality by construction (reorder.h): renumber variables and the rows of the zero / nonnegative cones when that makes the
    // gathers of the two CSR products share cache lines; everything below works in the new numbering
    plan_reorder(w->A, &w->k, w->has_P, w->reord);
    if (w->reord.active) apply_reorder(w->A, w->reord);
    phase("reorder");
    HIP_CHECK(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    phase("copy + stream");
    // equilibrate on the host copy (normalize_a_p) or identity scaling
    CsrPattern a_pattern; // filled by the device equilibration, reused by the linsys init
    if (w->stgs.normalize) {
      // 26 streaming passes over nnz: on the device once the matrix is big enough for
      // the ~200 launches to be cheaper than the host loops (SCS_AMD_EQUIL=host|dev forces)
      bool dev_eq = (long long)d->A->p[n] >= 100000;
      if (const char *e = opt_get("equil")) dev_eq = strcmp(e, "host") != 0;
      if (dev_eq) equilibrate_dev(w->has_P ? &w->P : nullptr, w->A, &w->k, w->scal, w->stream, &a_pattern);
      else equilibrate(w->has_P ? &w->P : nullptr, w->A, &w->k, w->scal);
    } else {
      w->scal.D.assign(m, (real)1);
      w->scal.E.assign(n, (real)1);
      w->scal.primal_scale = w->scal.dual_scale = 1;
    }
    phase("equilibrate");
    // device vectors
    for (DevBuf<real> *v : {&w->u, &w->u_t, &w->v, &w->v_prev, &w->rsk, &w->diag_r}) v->alloc(l);
    w->g.alloc(l - 1);
    w->b.alloc(m);
    w->c.alloc(n);
    w->D.alloc(m);
    w->E.alloc(n);
    w->warm.alloc(n);
    w->cw.alloc(m);
    w->ax.alloc(m);
    w->aty.alloc(n);
    if (w->has_P) w->px.alloc(n);
    w->part.alloc((size_t)NQ * PSTRIDE);
    w->qout.alloc(NQ);
    w->hq.alloc(NQ);
    w->D.upload(w->scal.D.data(), m, w->stream);
    w->E.upload(w->scal.E.data(), n, w->stream);
    w->b_orig.assign(d->b, d->b + m);
    w->c_orig.assign(d->c, d->c + n);
    if (w->reord.active) {
      for (int i = 0; i < m; ++i) w->b_orig[i] = d->b[w->reord.row_new2old[i]];
      for (int j = 0; j < n; ++j) w->c_orig[j] = d->c[w->reord.col_new2old[j]];
    }
    HIP_CHECK(hipStreamSynchronize(w->stream));
    if (scs_update(w, w->b_orig.data(), w->c_orig.data()) != 0) throw HipError("scs_amd: scs_update failed");
    phase("vectors + b,c");
    // linear system + cones on the device
    CscView Av = w->A.view(), Pv{nullptr, nullptr, nullptr, 0, 0};
    if (w->has_P) Pv = w->P.view();
    w->ls.init(&Av, w->has_P ? &Pv : nullptr, w->stream, &a_pattern);
    a_pattern.clear();
    phase("linsys init");
    set_diag_r(w);
    w->ls.set_diag_r_dev(w->diag_r.p);
    w->cone.init(&w->k, m, w->stgs.normalize ? w->scal.D.data() : nullptr, w->stream);
    if (w->stgs.acceleration_lookback) {
      // The O(l mem^2) work of AA runs on the device once l is large enough for the
      // PCIe round trip of v / v_prev and the host QR to matter; tiny problems keep the
      // host path (a handful of stream syncs would cost more than the arithmetic).
      // SCS_AMD_AA=host|dev forces either.
      bool dev_aa = l >= 32768;
      if (const char *e = opt_get("aa")) dev_aa = strcmp(e, "host") != 0;
      if (dev_aa) {
        w->accel_dev = aa_dev_init(l, w->stgs.acceleration_lookback, w->stgs.acceleration_lookback,
                                   w->stgs.acceleration_type_1, w->stgs.acceleration_regularization,
                                   w->stgs.acceleration_relaxation, (real)1., (real)1e10, 5, w->stream);
      } else {
        w->accel = aa_host_init(l, w->stgs.acceleration_lookback, w->stgs.acceleration_lookback,
                                w->stgs.acceleration_type_1, w->stgs.acceleration_regularization,
                                w->stgs.acceleration_relaxation, (real)1., (real)1e10, 5);
        if (w->accel) {
          w->hv.resize(l);
          w->hv_prev.resize(l);
        }
      }
      if (!w->accel && !w->accel_dev && w->stgs.verbose)
        printf("WARN: aa_init returned NULL, no acceleration applied.\n");
    }
    HIP_CHECK(hipStreamSynchronize(w->stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    printf("ERROR: init_work failure\n");
    delete w;
    return nullptr;
  }
  w->setup_time = now_ms() - t0;
  return w;
}

// The solve is split in three so that instrumentation (bench.py, trajectory tests)
// can time / inspect an exact range of iterations; scs_solve is begin + steps + end.
static void solve_begin(ScsWork *w, const ScsSolution *sol, scs_int warm_start) {
  const int n = w->n, m = w->m, l = w->l;
  HIP_CHECK(hipSetDevice(w->device)); // callers may be worker threads
  hipStream_t st = w->stream;
  w->t_solve0 = now_ms();
  w->t_lin = w->t_accel = 0;
  w->cone_timer.total_ms = 0;
  w->cone_timer.samples = 0;
  w->cur_iter = 0;
  w->run_status = SCS_UNFINISHED;
  w->loop_done = false;
  w->stepped = false;
  w->stgs.warm_start = warm_start;
  // reset_tracking (:1131-1144)
  w->last_scale_update_iter = 0;
  w->sum_log_scale_factor = 0;
  w->n_log_scale_factor = 0;
  w->scale_updates = 0;
  w->time_limit_reached = 0;
  w->rejected_accel_steps = w->accepted_accel_steps = 0;
  w->aa_norm = 0;
  w->r_n = Resid();
  w->r_o = Resid();
  if (!w->log_csv_name.empty()) { // open_csv_log_file, src/rw.c:686-698 (rewritten by every solve)
    if (w->log_csv_fout) fclose(w->log_csv_fout);
    w->log_csv_fout = fopen(w->log_csv_name.c_str(), "w");
    if (!w->log_csv_fout) printf("Error: Could not open %s for writing\n", w->log_csv_name.c_str());
    else fputs(LOG_CSV_HEADER, w->log_csv_fout);
  }
  // warm / cold start (:660-687)
  std::vector<real> hv(l, (real)0);
  if (warm_start && sol && sol->x && sol->y && sol->s) {
    std::vector<real> x(sol->x, sol->x + n), y(sol->y, sol->y + m), s(sol->s, sol->s + m);
    if (w->reord.active) { // the caller's numbering -> ours
      for (int j = 0; j < n; ++j) x[j] = sol->x[w->reord.col_new2old[j]];
      for (int i = 0; i < m; ++i) {
        y[i] = sol->y[w->reord.row_new2old[i]];
        s[i] = sol->s[w->reord.row_new2old[i]];
      }
    }
    if (w->stgs.normalize) normalize_sol(w->scal, x.data(), y.data(), s.data());
    std::vector<real> hr(l);
    w->diag_r.download(hr.data(), l, st);
    HIP_CHECK(hipStreamSynchronize(st));
    for (int j = 0; j < n; ++j) hv[j] = x[j] != x[j] ? (real)0 : x[j];
    for (int j = 0; j < m; ++j) {
      real t = y[j] + s[j] / hr[n + j];
      hv[n + j] = t != t ? (real)0 : t;
    }
  }
  hv[l - 1] = 1;
  w->v.upload(hv.data(), l, st);
  HIP_CHECK(hipMemsetAsync(w->u.p, 0, l * sizeof(real), st));
  HIP_CHECK(hipMemsetAsync(w->u_t.p, 0, l * sizeof(real), st));
  HIP_CHECK(hipMemsetAsync(w->rsk.p, 0, l * sizeof(real), st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (w->accel) aa_host_reset(w->accel);
  if (w->accel_dev) aa_dev_reset(w->accel_dev);
  w->cone.reset_warm_start(); // every solve starts its PSD eigenbases cold: reruns are bit-identical
  update_work_cache(w);
  if (w->stgs.verbose) print_header(w);
}

// run iterations [cur_iter, upto) of the loop of src/scs.c:1356-1455; stops early on
// convergence / time limit (loop_done).  Returns <0 on failure.
static int solve_steps(ScsWork *w, int upto) {
  const int n = w->n, m = w->m, l = w->l;
  HIP_CHECK(hipSetDevice(w->device));
  hipStream_t st = w->stream;
  const int gl = glue_grid(l), gnm = glue_grid(n + m);
  real *rp_part = w->part.p; // root_plus partials reuse the residual partial area (5 rows)
  int &i = w->cur_iter;
  for (; i < upto && !w->loop_done; ++i) {
    // ---- Anderson acceleration (host) :1359-1366
    if (w->accel_dev) {
      const double ta = now_ms();
      if (i > 0 && i % w->stgs.acceleration_interval == 0) {
        w->aa_norm = aa_dev_apply(w->v.p, w->v_prev.p, w->accel_dev);
        HIP_CHECK(hipStreamSynchronize(st)); // so that accel_time is the AA's own time
      }
      w->t_accel += now_ms() - ta;
    } else if (w->accel) {
      const double ta = now_ms();
      if (i > 0 && i % w->stgs.acceleration_interval == 0) {
        w->v.download(w->hv.data(), l, st);
        w->v_prev.download(w->hv_prev.data(), l, st);
        HIP_CHECK(hipStreamSynchronize(st));
        w->aa_norm = aa_host_apply(w->hv.data(), w->hv_prev.data(), w->accel);
        w->v.upload(w->hv.data(), l, st);
        HIP_CHECK(hipStreamSynchronize(st));
      }
      w->t_accel += now_ms() - ta;
    }
    // ---- normalize v, v_prev, u_t, warm start :1368-1377, :738-758
    const int do_norm = i >= FEASIBLE_ITERS;
    if (do_norm)
      hipLaunchKernelGGL(k_sumsq_partial, dim3(gl), dim3(SCSAMD_BLOCK), 0, st, w->v.p, l, w->part.p + 8 * PSTRIDE);
    hipLaunchKernelGGL(k_prep_linsys, dim3(gl), dim3(SCSAMD_BLOCK), 0, st, w->v.p,
                       (w->accel || w->accel_dev) ? w->v_prev.p : (real *)nullptr, w->u_t.p, w->u.p, w->g.p, w->diag_r.p,
                       w->warm.p, n, l, w->part.p + 8 * PSTRIDE, gl, w->part.p + 9 * PSTRIDE, do_norm);
    // ---- linear system :763 with the tolerance schedule of :745-762
    {
      const double tl = now_ms();
      const real warm_scale = (real)1.0 / std::pow((real)i + 1, (real)CG_RATE);
      if (w->cg_tol_override > 0) {
        w->ls.solve_dev(w->u_t.p, w->warm.p, (real)w->cg_tol_override);
      } else {
        const real tol_cap = std::min(w->r_n.nm_ax_s_btau, w->r_n.nm_px_aty_ctau);
        w->ls.solve_dev(w->u_t.p, w->warm.p, tol_cap, w->part.p + 9 * PSTRIDE, gl, warm_scale);
      }
      w->t_lin += now_ms() - tl;
    }
    // ---- tau~, u_t, u = 2 u_t - v, Moreau pre :764-769, :796-800
    const int feas = i < FEASIBLE_ITERS;
    if (!feas)
      hipLaunchKernelGGL(k_root_plus_partial, dim3(gnm), dim3(SCSAMD_BLOCK), 0, st, w->u_t.p, w->v.p, w->g.p,
                         w->diag_r.p, n + m, rp_part, PSTRIDE);
    hipLaunchKernelGGL(k_post_linsys, dim3(gl), dim3(SCSAMD_BLOCK), 0, st, w->u_t.p, w->u.p, w->v.p, w->g.p,
                       w->diag_r.p, w->cw.p, n, l, rp_part, gnm, PSTRIDE, feas);
    // ---- cone projection :803
    int cslot = -1;
    if (w->cone_timer.used < 500) cslot = w->cone_timer.start(st);
    w->cone.proj_primal(w->cw.p, w->diag_r.p + n);
    w->cone_timer.stop(cslot, st);
    w->cone_projs++;
    // ---- rsk (+ dual update when nothing can intervene) :1397, :1432
    const bool check = i % CONVERGED_INTERVAL == 0;
    const bool print = w->stgs.verbose && i % PRINT_INTERVAL == 0;
    const bool fuse_dual = !check && !print;
    hipLaunchKernelGGL(k_post_cone, dim3(gl), dim3(SCSAMD_BLOCK), 0, st, w->u.p, w->u_t.p, w->v.p, w->rsk.p,
                       w->diag_r.p, w->cw.p, n, l, fuse_dual ? w->stgs.alpha : (real)0);
    if (check) {
      if (interrupted()) return -2; // :1400-1403
      populate_residuals(w, i);
      if ((w->run_status = has_converged(w)) != 0) {
        w->loop_done = true;
        break; // like the reference, the converged iteration is not counted (:1405-1407)
      }
      if (w->stgs.time_limit_secs && now_ms() - w->t_solve0 > 1000. * w->stgs.time_limit_secs) {
        w->time_limit_reached = 1;
        w->loop_done = true;
        break;
      }
    }
    if (print) {
      populate_residuals(w, i);
      print_summary(w, i, w->t_solve0);
    }
    if (w->stgs.adaptive_scale && i == w->r_o.last_iter) {
      if (update_scale(w, i) < 0) return -1;
    }
    if (!fuse_dual)
      hipLaunchKernelGGL(k_dual_update, dim3(gl), dim3(SCSAMD_BLOCK), 0, st, w->v.p, w->u.p, w->u_t.p, l,
                         w->stgs.alpha);
    // ---- AA safeguard :1439-1447
    if (w->accel_dev && i % w->stgs.acceleration_interval == 0 && w->aa_norm > 0) {
      const double ta = now_ms();
      if (aa_dev_safeguard(w->v.p, w->v_prev.p, w->accel_dev) < 0) w->rejected_accel_steps++;
      else w->accepted_accel_steps++;
      w->t_accel += now_ms() - ta;
    } else if (w->accel && i % w->stgs.acceleration_interval == 0 && w->aa_norm > 0) {
      const double ta = now_ms();
      w->v.download(w->hv.data(), l, st);
      w->v_prev.download(w->hv_prev.data(), l, st);
      HIP_CHECK(hipStreamSynchronize(st));
      if (aa_host_safeguard(w->hv.data(), w->hv_prev.data(), w->accel) < 0) {
        w->rejected_accel_steps++;
        w->v.upload(w->hv.data(), l, st);
        w->v_prev.upload(w->hv_prev.data(), l, st);
        HIP_CHECK(hipStreamSynchronize(st));
      } else {
        w->accepted_accel_steps++;
      }
      w->t_accel += now_ms() - ta;
    }
    // log after the scale update so that the residual recomputation does not change the
    // algorithm's own cadence more than the reference's does (:1449-1454)
    if (w->log_csv_fout) log_csv_row(w, i);
    else if (w->resid_every_iter) populate_residuals(w, i);
  }
  return 0;
}

static void solve_end(ScsWork *w, ScsSolution *sol, ScsInfo *info) {
  const int i = w->cur_iter;
  HIP_CHECK(hipSetDevice(w->device));
  strcpy(info->lin_sys_solver, scs_get_lin_sys_method());
  info->status_val = w->run_status;
  if (w->log_csv_fout) { // final row + close (:1457-1461, :1481)
    log_csv_row(w, i);
    fclose(w->log_csv_fout);
    w->log_csv_fout = nullptr;
  }
  if (w->stgs.verbose) {
    populate_residuals(w, i);
    print_summary(w, i, w->t_solve0);
  }
  finalize(w, sol, info, i);
  HIP_CHECK(hipGetLastError());
  w->cone_timer.harvest();
  info->solve_time = (real)(now_ms() - w->t_solve0);
  info->lin_sys_time = (real)w->t_lin;
  info->cone_time = (real)(w->cone_timer.samples
                               ? w->cone_timer.total_ms * ((double)std::max(i, 1) / (double)w->cone_timer.samples)
                               : 0.0);
  info->accel_time = (real)w->t_accel;
  if (w->stgs.verbose) print_footer(info);
}

scs_int scs_solve(ScsWork *w, ScsSolution *sol, ScsInfo *info, scs_int warm_start) { // :1327-1484
  if (!sol || !w || !info) {
    printf("ERROR: missing ScsWork, ScsSolution or ScsInfo input\n");
    return SCS_FAILED;
  }
  InterruptListener listener; // :1344, restored on every return below (:369, :1482)
  try {
    solve_begin(w, sol, warm_start);
    const int rc = solve_steps(w, w->stgs.max_iters);
    if (rc < 0) {
      if (w->log_csv_fout) {
        fclose(w->log_csv_fout);
        w->log_csv_fout = nullptr;
      }
      HIP_CHECK(hipStreamSynchronize(w->stream));
      if (rc == -2) return fail_out(w, w->m, w->n, sol, info, SCS_SIGINT, "interrupted", "interrupted");
      return fail_out(w, w->m, w->n, sol, info, SCS_FAILED, "error in update_scale", "failure");
    }
    solve_end(w, sol, info);
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return fail_out(w, w->m, w->n, sol, info, SCS_FAILED, "HIP error in scs_solve", "failure");
  }
  return info->status_val;
}

// ---- instrumentation: the same solve in three calls (not in the reference) ------
scs_int scs_amd_solve_begin(ScsWork *w, const ScsSolution *sol, scs_int warm_start) {
  if (!w) return -1;
  try {
    solve_begin(w, sol, warm_start);
    HIP_CHECK(hipStreamSynchronize(w->stream));
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return 0;
}
// Runs up to `steps` further ADMM iterations; returns the iteration counter
// afterwards (it stops advancing once converged), or <0 on failure.  The stream is
// idle on return.
scs_int scs_amd_solve_steps(ScsWork *w, scs_int steps) {
  if (!w) return -1;
  try {
    w->stepped = true;
    const long long upto = std::min<long long>((long long)w->cur_iter + steps, w->stgs.max_iters);
    if (solve_steps(w, (int)upto) < 0) return -1;
    HIP_CHECK(hipStreamSynchronize(w->stream));
    HIP_CHECK(hipGetLastError());
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
  return w->cur_iter;
}
scs_int scs_amd_solve_converged(const ScsWork *w) { return w ? (w->loop_done ? 1 : 0) : -1; }
scs_int scs_amd_solve_end(ScsWork *w, ScsSolution *sol, ScsInfo *info) {
  if (!w || !sol || !info) return SCS_FAILED;
  try {
    solve_end(w, sol, info);
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return fail_out(w, w->m, w->n, sol, info, SCS_FAILED, "HIP error in scs_solve", "failure");
  }
  return info->status_val;
}
// test hook: force every per-iteration linear solve to this tolerance (0 = schedule)
void scs_amd_set_cg_tol_override(ScsWork *w, double tol) {
  if (w) w->cg_tol_override = tol;
}

// measurement hook: refresh the residuals after EVERY iteration, at the place where the reference does so when
// `log_csv_filename` is set (src/scs.c:1449-1454).  The refreshed normalised norms feed the CG tolerance of the next
// iteration (src/scs.c:745-762), so a logged reference run follows a different (tighter) tolerance schedule than an
// unlogged one; this switch puts our solve on the logged schedule without the host-side CSV work, so that bench.py
// can time the same schedule the reference's logged CPU window ran.
void scs_amd_set_residuals_every_iter(ScsWork *w, scs_int on) {
  if (w) w->resid_every_iter = on != 0;
}

// what scs_init decided about the internal numbering (reorder.h): out[0] = 1 if variables / rows were renumbered, out[1..2] =
// distinct 128-byte lines per gathered entry of the A and A' products as given, out[3..4] = the same after the renumbering
// (equal to before when none was tried), out[5] = seconds spent deciding
void scs_amd_get_reorder_info(const ScsWork *w, double *out) {
  if (!w || !out) return;
  out[0] = w->reord.active ? 1 : 0;
  out[1] = w->reord.before[0];
  out[2] = w->reord.before[1];
  out[3] = w->reord.after[0];
  out[4] = w->reord.after[1];
  out[5] = w->reord.seconds;
}

// how scs_init laid the matrices out (spmv_wave.h / spmv_wave_build.h): for A then A': out[0], out[3] = wave-owned-rows layout built
// (0 / 1), out[1], out[4] = built on the device (0 / 1), out[2], out[5] = distinct 128-byte lines per gathered entry
void scs_amd_get_layout_info(const ScsWork *w, double *out) {
  if (!w || !out) return;
  const WaveRowsDev *wv[2] = {w->ls.A.wave, w->ls.At.wave};
  for (int i = 0; i < 2; ++i) {
    const bool b = wv[i] && wv[i]->built;
    out[3 * i] = b ? 1 : 0;
    out[3 * i + 1] = b && wv[i]->built_on_device ? 1 : 0;
    out[3 * i + 2] = b ? wv[i]->lines_per_entry : 0;
  }
}

// which SpMV kernel scs_init chose for A (which = 0) / A' (which = 1), as the template instantiation's name (what rocprofv3 lists):
// bench.py labels its roofline block with it (the label was hard-coded).  Returns the length needed.
scs_int scs_amd_get_spmv_kernel_name(const ScsWork *w, scs_int which, char *buf, scs_int cap) {
  if (!w) return -1;
  const WaveRowsDev *wv = which ? w->ls.At.wave : w->ls.A.wave;
  char tmp[96];
  if (wv && wv->built && wv->wide) snprintf(tmp, sizeof tmp, "csr_wave_wide_kernel<EPI>");
  else if (wv && wv->built && wv->lockstep) snprintf(tmp, sizeof tmp, "csr_wave_lockstep_kernel<EPI,%d,%d>", wv->ls_wpb, wv->ls_bmode);
  else if (wv && wv->built) snprintf(tmp, sizeof tmp, "csr_wave_kernel<EPI,%d>", wv->pipelined);
  else snprintf(tmp, sizeof tmp, "csr_stream_kernel<EPI>");
  const size_t len = strlen(tmp);
  if (buf && cap > 0) {
    const size_t c = std::min(len, (size_t)cap - 1);
    memcpy(buf, tmp, c);
    buf[c] = 0;
  }
  return (scs_int)len;
}

// test hook (host only, no HIP call): the decision of reorder.h on a caller's matrix.  col_new2old (n) / row_new2old (m) receive the
// numbering (identity when none is kept); info as scs_amd_get_reorder_info.  Returns 1 if a renumbering was kept, 0 if not, <0 on error.
scs_int scs_amd_plan_reorder(const ScsMatrix *A, const ScsCone *k, scs_int *col_new2old, scs_int *row_new2old, double *info) {
  if (!A || !k || !col_new2old || !row_new2old) return -1;
  if (k->z < 0 || k->l < 0 || (long long)k->z + k->l > (long long)A->m) return -1; // plan_reorder indexes rows [0, z + l)
  try {
    HostCsc a;
    a.copy_from(A);
    Reorder R;
    plan_reorder(a, k, false, R);
    for (scs_int j = 0; j < A->n; ++j) col_new2old[j] = R.active ? (scs_int)R.col_new2old[j] : j;
    for (scs_int i = 0; i < A->m; ++i) row_new2old[i] = R.active ? (scs_int)R.row_new2old[i] : i;
    if (info) {
      info[0] = R.active ? 1 : 0;
      info[1] = R.before[0];
      info[2] = R.before[1];
      info[3] = R.after[0];
      info[4] = R.after[1];
      info[5] = R.seconds;
    }
    return R.active ? 1 : 0;
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
}

// test hook: the equilibration of scs_init on caller-owned arrays, host or device
scs_int scs_amd_equilibrate(ScsMatrix *A, ScsMatrix *P, const ScsCone *k, scs_float *D, scs_float *E,
                            scs_int where) {
  try {
    HostCsc a, p;
    a.copy_from(A);
    if (P) p.copy_from(P);
    Scaling sc;
    if (where) {
      if (scs_amd_device_count() <= 0) throw HipError("scs_amd: no HIP device visible");
      HIP_CHECK(hipSetDevice(selected_device()));
      hipStream_t st;
      HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      try {
        equilibrate_dev(P ? &p : nullptr, a, k, sc, st, nullptr);
      } catch (...) {
        (void)hipStreamDestroy(st);
        throw;
      }
      (void)hipStreamDestroy(st);
    } else {
      equilibrate(P ? &p : nullptr, a, k, sc);
    }
    memcpy(A->x, a.x.data(), a.x.size() * sizeof(real));
    if (P) memcpy(P->x, p.x.data(), p.x.size() * sizeof(real));
    memcpy(D, sc.D.data(), sc.D.size() * sizeof(real));
    memcpy(E, sc.E.data(), sc.E.size() * sizeof(real));
    return 0;
  } catch (const std::exception &ex) {
    fprintf(stderr, "%s\n", ex.what());
    return -1;
  }
}

void scs_finish(ScsWork *w) {
  if (!w) return;
  (void)hipSetDevice(w->device);
  delete w;
}

scs_int scs(const ScsData *d, const ScsCone *k, const ScsSettings *stgs, ScsSolution *sol,
            ScsInfo *info) { // :1538-1551
  scs_int status;
  ScsWork *w = scs_init(d, k, stgs);
  if (w) {
    scs_solve(w, sol, info, stgs->warm_start);
    status = info->status_val;
  } else {
    status = fail_out(nullptr, d ? d->m : -1, d ? d->n : -1, sol, info, SCS_FAILED, "could not initialize work",
                      "failure");
  }
  scs_finish(w);
  return status;
}

void scs_amd_set_profiling(ScsWork *w, scs_int on) {
  if (!w) return;
  w->profiling = on != 0;
  w->ls.profiling = on != 0;
}

void scs_amd_get_stats(const ScsWork *cw, ScsAmdStats *out) {
  if (!cw || !out) return;
  ScsWork *w = const_cast<ScsWork *>(cw);
  (void)hipSetDevice(w->device);
  (void)hipStreamSynchronize(w->stream);
  w->ls.harvest_timers();
  memset(out, 0, sizeof *out);
  out->cg_iters = w->ls.tot_cg_its;
  out->lin_sys_solves = w->ls.n_solves;
  out->mat_vecs = w->ls.n_matvecs;
  out->spmv_launches = w->ls.spmv_timer.samples;
  out->spmv_ms = w->ls.spmv_timer.total_ms;
  out->cg_ms = w->ls.cg_timer.total_ms;
  out->cone_ms = w->cone_timer.total_ms;
  out->cone_projs = w->cone_timer.samples;
  out->nnz = w->ls.A.nnz;
  out->spmv_bytes = w->ls.matvec_bytes();
  out->psd_unconverged = w->psd_unconverged;
}

} // extern "C"
