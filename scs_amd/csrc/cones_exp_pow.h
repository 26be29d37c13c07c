// cones_exp_pow.h -- exponential and power cone projections, one lane per 3-row cone.
//
// SURVEY.md section 8(f) item 3 (the first "next" row after the hot path): these cones
// appear in CVXPY-generated problems and are embarrassingly parallel (the reference
// itself OpenMPs the exponential loop, src/cones.c:1407-1412).  What is computed follows
//   exponential cone: src/exp_cone.c:373-441 `proj_pd_exp_cone` -- Friberg (2021),
//       "Projection onto the exponential cone: a univariate root-finding problem":
//       heuristic projections on primal/polar (:160-207), optimality shortcut (:395-411),
//       bracket for the root of h (:249-317), damped Newton with bisection fallback
//       (:66-157), conversion of the root to the primal/polar point (:320-367);
//   power cone: src/cones.c:1284-1335 `proj_power_cone` (Newton on r, <= 20 steps) and
//       the Moreau form for dual power cones (:1427-1441).
// All scalar work is done in the build's scs_float, with the same constants and the same
// branch order, so results match the reference to rounding.
#pragma once
#include "common.h"

namespace scsamd {

#ifdef __HIPCC__
typedef scs_float ereal;

#define EXPC_INF ((ereal)1e15)

__device__ __forceinline__ bool ec_finite(ereal x) { return absval(x) < EXPC_INF; }
__device__ __forceinline__ ereal ec_max(ereal a, ereal b) { return a > b ? a : b; }
__device__ __forceinline__ ereal ec_min(ereal a, ereal b) { return a < b ? a : b; }
__device__ __forceinline__ ereal ec_clip(ereal x, ereal l, ereal u) { return ec_max(l, ec_min(u, x)); }
__device__ __forceinline__ ereal ec_safediv_pos(ereal x, ereal y) { return y < (ereal)1e-18 ? x / (ereal)1e-18 : x / y; }
__device__ __forceinline__ ereal ec_dist_sq(const ereal *a, const ereal *b) {
  const ereal d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  return d0 * d0 + d1 * d1 + d2 * d2;
}

// h(rho) and h'(rho), exp_cone.c:41-64
__device__ __forceinline__ void ec_h(const ereal *v0, ereal rho, ereal *f, ereal *df) {
  const ereal t0 = v0[2], s0 = v0[1], r0 = v0[0];
  const ereal er = exp(rho), enr = (ereal)1.0 / er;
  *f = ((rho - 1) * r0 + s0) * er - (r0 - rho * s0) * enr - (rho * (rho - 1) + 1) * t0;
  if (df) *df = (rho * r0 + s0) * er + (r0 - (rho - 1) * s0) * enr - (2 * rho - 1) * t0;
}

__device__ ereal ec_bisect(const ereal *v0, ereal xl, ereal xu, ereal x) { // exp_cone.c:67-98
  ereal x_plus = x, f;
  for (int i = 0; i < 40; ++i) {
    ec_h(v0, x, &f, nullptr);
    if (f < (ereal)0.0) xl = x;
    else xu = x;
    x_plus = (ereal)0.5 * (xl + xu);
    if (absval(x_plus - x) <= (ereal)1e-12 * ec_max((ereal)1.0, absval(x_plus)) || x_plus == xl || x_plus == xu) break;
    x = x_plus;
  }
  return x_plus;
}

__device__ ereal ec_newton(const ereal *v0, ereal xl, ereal xu, ereal x) { // exp_cone.c:101-157
  const ereal EPS = (ereal)1e-15, DFTOL = (ereal)1e-13, LODAMP = (ereal)0.05, HIDAMP = (ereal)0.95;
  ereal x_plus, f, df;
  int i;
  for (i = 0; i < 20; ++i) {
    ec_h(v0, x, &f, &df);
    if (absval(f) <= EPS) break;
    if (f < (ereal)0.0) xl = x;
    else xu = x;
    if (xu <= xl) {
      xu = (ereal)0.5 * (xu + xl);
      xl = xu;
      break;
    }
    if (!ec_finite(f) || df < DFTOL) break;
    x_plus = x - f / df;
    if (absval(x_plus - x) <= EPS * ec_max((ereal)1.0, absval(x_plus))) break;
    if (x_plus >= xu) x = ec_min(LODAMP * x + HIDAMP * xu, xu);
    else if (x_plus <= xl) x = ec_max(LODAMP * x + HIDAMP * xl, xl);
    else x = x_plus;
  }
  if (i < 20) return ec_clip(x, xl, xu);
  return ec_bisect(v0, xl, xu, x);
}

__device__ ereal ec_heur_primal(const ereal *v0, ereal *vp) { // exp_cone.c:160-182
  const ereal t0 = v0[2], s0 = v0[1], r0 = v0[0];
  vp[2] = ec_max(t0, (ereal)0.0);
  vp[1] = 0;
  vp[0] = ec_min(r0, (ereal)0.0);
  ereal d = ec_dist_sq(v0, vp);
  if (s0 > (ereal)0.0) {
    const ereal tp = ec_max(t0, s0 * exp(r0 / s0));
    const ereal nd = (tp - t0) * (tp - t0);
    if (nd < d) {
      vp[2] = tp;
      vp[1] = s0;
      vp[0] = r0;
      d = nd;
    }
  }
  return d;
}
__device__ ereal ec_heur_polar(const ereal *v0, ereal *vd) { // exp_cone.c:185-207
  const ereal t0 = v0[2], s0 = v0[1], r0 = v0[0];
  vd[2] = ec_min(t0, (ereal)0.0);
  vd[1] = ec_min(s0, (ereal)0.0);
  vd[0] = 0;
  ereal d = ec_dist_sq(v0, vd);
  if (r0 > (ereal)0.0) {
    const ereal td = ec_min(t0, -r0 * exp(s0 / r0 - (ereal)1.0));
    const ereal nd = (t0 - td) * (t0 - td);
    if (nd < d) {
      vd[2] = td;
      vd[1] = s0;
      vd[0] = r0;
      d = nd;
    }
  }
  return d;
}

__device__ ereal ec_ppsi(const ereal *v0) { // exp_cone.c:209-220
  const ereal s0 = v0[1], r0 = v0[0];
  const ereal q = sqrt(r0 * r0 + s0 * s0 - r0 * s0);
  const ereal psi = r0 > s0 ? (r0 - s0 + q) / r0 : -s0 / (r0 - s0 - q);
  return ((psi - (ereal)1.0) * r0 + s0) / (psi * (psi - (ereal)1.0) + (ereal)1.0);
}
__device__ ereal ec_pomega(ereal rho) { // :222-229
  ereal v = exp(rho) / (rho * (rho - (ereal)1.0) + (ereal)1.0);
  if (rho < (ereal)2.0) v = ec_min(v, exp((ereal)2.0) / (ereal)3.0);
  return v;
}
__device__ ereal ec_dpsi(const ereal *v0) { // :231-242
  const ereal s0 = v0[1], r0 = v0[0];
  const ereal q = sqrt(r0 * r0 + s0 * s0 - r0 * s0);
  const ereal psi = s0 > r0 ? (r0 - q) / s0 : (r0 - s0) / (r0 + q);
  return (r0 - psi * s0) / (psi * (psi - (ereal)1.0) + (ereal)1.0);
}
__device__ ereal ec_domega(ereal rho) { // :244-251
  ereal v = -exp(-rho) / (rho * (rho - (ereal)1.0) + (ereal)1.0);
  if (rho > (ereal)-1.0) v = ec_max(v, -exp((ereal)1.0) / (ereal)3.0);
  return v;
}

__device__ void ec_bracket(const ereal *v0, ereal pd, ereal dd, ereal *lo, ereal *up) { // exp_cone.c:254-317
  const ereal t0 = v0[2], s0 = v0[1], r0 = v0[0];
  ereal baselow = -EXPC_INF, baseupr = EXPC_INF, low = -EXPC_INF, upr = EXPC_INF;
  const ereal ms = ec_min(s0, (ereal)0.0), mr = ec_min(r0, (ereal)0.0);
  const ereal Dp = sqrt(ec_max(pd - ms * ms, (ereal)0.0)), Dd = sqrt(ec_max(dd - mr * mr, (ereal)0.0));
  if (t0 > (ereal)0.0) low = ec_max(low, log(t0 / ec_ppsi(v0)));
  else if (t0 < (ereal)0.0) upr = ec_min(upr, -log(-t0 / ec_dpsi(v0)));
  if (r0 > (ereal)0.0) {
    baselow = (ereal)1.0 - s0 / r0;
    low = ec_max(low, baselow);
    const ereal tpu = ec_max((ereal)1e-12, ec_min(Dd, Dp + t0));
    const ereal val = r0 * ec_pomega(low);
    const ereal sgn = val < 0 ? (ereal)-1 : (ereal)1;
    upr = ec_min(upr, ec_max(low, baselow + ec_safediv_pos(tpu, absval(val)) * sgn));
  }
  if (s0 > (ereal)0.0) {
    baseupr = r0 / s0;
    upr = ec_min(upr, baseupr);
    const ereal tdl = -ec_max((ereal)1e-12, ec_min(Dp, Dd - t0));
    const ereal val = s0 * ec_domega(upr);
    const ereal sgn = val < 0 ? (ereal)-1 : (ereal)1;
    low = ec_max(low, ec_min(upr, baseupr - ec_safediv_pos(tdl, absval(val)) * sgn));
  }
  low = ec_clip(ec_min(low, upr), baselow, baseupr);
  upr = ec_clip(ec_max(low, upr), baselow, baseupr);
  if (low != upr) {
    ereal fl, fu;
    ec_h(v0, low, &fl, nullptr);
    ec_h(v0, upr, &fu, nullptr);
    if (fl * fu > (ereal)0.0) {
      if (absval(fl) < absval(fu)) upr = low;
      else low = upr;
    }
  }
  *lo = low;
  *up = upr;
}

__device__ ereal ec_sol_primal(const ereal *v0, ereal rho, ereal *vp) { // exp_cone.c:320-342
  const ereal lin = (rho - (ereal)1.0) * v0[0] + v0[1], er = exp(rho);
  if (lin > (ereal)0.0 && ec_finite(er)) {
    const ereal q = rho * (rho - (ereal)1.0) + (ereal)1.0;
    vp[2] = er * lin / q;
    vp[1] = lin / q;
    vp[0] = rho * lin / q;
    return ec_dist_sq(vp, v0);
  }
  vp[2] = EXPC_INF;
  vp[1] = 0;
  vp[0] = 0;
  return EXPC_INF;
}
__device__ ereal ec_sol_polar(const ereal *v0, ereal rho, ereal *vd) { // exp_cone.c:345-367
  const ereal lin = v0[0] - rho * v0[1], er = exp(-rho);
  if (lin > (ereal)0.0 && ec_finite(er)) {
    const ereal q = rho * (rho - (ereal)1.0) + (ereal)1.0;
    vd[2] = -er * lin / q;
    vd[1] = ((ereal)1.0 - rho) * lin / q;
    vd[0] = lin / q;
    return ec_dist_sq(v0, vd);
  }
  vd[2] = -EXPC_INF;
  vd[1] = 0;
  vd[0] = 0;
  return EXPC_INF;
}

// in-place projection of one triple onto the exponential cone (primal != 0) or its dual
__device__ void proj_exp_cone_triple(ereal *v0, int primal) { // exp_cone.c:373-441
  const ereal TOL = (ereal)1e-8;
  ereal vp[3], vd[3], vh[3], xl, xh;
  if (!primal) {
    v0[0] = -v0[0];
    v0[1] = -v0[1];
    v0[2] = -v0[2];
  }
  ereal pd = ec_heur_primal(v0, vp), dd = ec_heur_polar(v0, vd);
  ereal err = absval(vp[0] + vd[0] - v0[0]);
  err = ec_max(err, absval(vp[1] + vd[1] - v0[1]));
  err = ec_max(err, absval(vp[2] + vd[2] - v0[2]));
  bool opt = v0[1] <= (ereal)0.0 && v0[0] <= (ereal)0.0;
  opt = opt || ec_min(pd, dd) <= TOL * TOL;
  opt = opt || (err <= TOL && (vp[0] * vd[0] + vp[1] * vd[1] + vp[2] * vd[2]) <= TOL);
  if (!opt) {
    ec_bracket(v0, pd, dd, &xl, &xh);
    const ereal rho = ec_newton(v0, xl, xh, (ereal)0.5 * (xl + xh));
    if (primal) {
      const ereal dh = ec_sol_primal(v0, rho, vh);
      if (dh <= pd) {
        vp[0] = vh[0];
        vp[1] = vh[1];
        vp[2] = vh[2];
      }
    } else {
      const ereal dh = ec_sol_polar(v0, rho, vh);
      if (dh <= dd) {
        vd[0] = vh[0];
        vd[1] = vh[1];
        vd[2] = vh[2];
      }
    }
  }
  if (primal) {
    v0[0] = vp[0];
    v0[1] = vp[1];
    v0[2] = vp[2];
  } else { // polar -> dual
    v0[0] = -vd[0];
    v0[1] = -vd[1];
    v0[2] = -vd[2];
  }
}

// ---- power cone -------------------------------------------------------------------------
__device__ __forceinline__ ereal pc_x(ereal r, ereal xh, ereal rh, ereal a) { // cones.c:1284-1288
  const ereal x = (ereal)0.5 * (xh + sqrt(xh * xh + 4 * a * (rh - r) * r));
  return ec_max(x, (ereal)1e-12);
}
__device__ void proj_power_cone_triple(ereal *v, ereal a) { // cones.c:1290-1335
  const ereal PTOL = (ereal)1e-9;
  const ereal xh = v[0], yh = v[1], rh = absval(v[2]);
  ereal x = 0, y = 0, r;
  if (xh >= 0 && yh >= 0 && PTOL + pow(xh, a) * pow(yh, (1 - a)) >= rh) return;
  if (xh <= 0 && yh <= 0 && PTOL + pow(-xh, a) * pow(-yh, 1 - a) >= rh * pow(a, a) * pow(1 - a, 1 - a)) {
    v[0] = v[1] = v[2] = 0;
    return;
  }
  r = rh / 2;
  for (int i = 0; i < 20; ++i) {
    x = pc_x(r, xh, rh, a);
    y = pc_x(r, yh, rh, 1 - a);
    const ereal xa = pow(x, a), y1a = pow(y, (1 - a));
    const ereal f = xa * y1a - r;
    if (absval(f) < PTOL) break;
    const ereal dxdr = a * (rh - 2 * r) / (2 * x - xh);
    const ereal dydr = (1 - a) * (rh - 2 * r) / (2 * y - yh);
    const ereal fp = xa * y1a * (a * dxdr / x + (1 - a) * dydr / y) - 1;
    r = ec_max(r - f / fp, (ereal)0);
    r = ec_min(r, rh);
  }
  v[0] = x;
  v[1] = y;
  v[2] = (v[2] < 0) ? -r : r;
}

// one lane per cone: [ep primal exp | ed dual exp | psize power (a<0 means dual)]
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_exp_pow(scs_float *x, int ep, int ed, int psize,
                                                          const scs_float *__restrict__ pw) {
  const int total = ep + ed + psize;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    scs_float *v = x + 3 * (size_t)c;
    ereal t[3] = {v[0], v[1], v[2]};
    if (c < ep + ed) {
      proj_exp_cone_triple(t, c < ep);
    } else {
      const ereal a = pw[c - ep - ed];
      if (a >= 0) {
        proj_power_cone_triple(t, a);
      } else { // dual power cone via Moreau, cones.c:1427-1441
        ereal w[3] = {-t[0], -t[1], -t[2]};
        proj_power_cone_triple(w, -a);
        t[0] += w[0];
        t[1] += w[1];
        t[2] += w[2];
      }
    }
    v[0] = t[0];
    v[1] = t[1];
    v[2] = t[2];
  }
}
#endif // __HIPCC__

} // namespace scsamd
