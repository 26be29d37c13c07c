// cones_exp_pow.h -- exponential / dual exponential / power / dual power cones (3 rows each) for CDNA4.
//
// What must come out is what the reference returns (src/exp_cone.c:373-441 `proj_pd_exp_cone`,
// src/cones.c:1290-1335 `proj_power_cone`, Moreau form for the dual power cone :1427-1441), to rounding.
// How it is computed here is laid out for 64-wide wavefronts instead of one scalar cone at a time:
//
//  * a 256-lane workgroup stages 256 consecutive triples (768 values) through LDS with unit-stride global loads and
//    stores; each lane then owns one triple in registers;
//  * control flow is wave-uniform: every lane evaluates the cheap closed-form candidates, the wave votes
//    (`wave_any`) on whether anybody needs the root search, and the search itself is a predicated loop -- lanes
//    that are finished freeze their state, the loop leaves when the whole wave is finished -- so no lane ever
//    waits in a divergent branch of its neighbour;
//  * the exponential-cone root search is a bracketed Newton iteration whose bracket only moves through
//    min / max selects (no data-dependent exits, no separate bisection routine).
//
// Exponential cone, the geometry used (u = v[0], w = v[1], t = v[2]; K = cl{w > 0, w e^(u/w) <= t}):
// the boundary of K is swept by the rays g(rho) = (rho, 1, e^rho), and the outward normal at g(rho) is
// n(rho) = (1, 1 - rho, -e^-rho), which sweeps the boundary of the polar cone.  A point outside both cones
// decomposes as v = a g(rho) + b n(rho) with a, b > 0 (Moreau); the first two rows give
//     a = ((rho - 1) u + w) / q,   b = (u - rho w) / q,   q = rho^2 - rho + 1,
// and the third row is one equation in rho alone,
//     F(rho) = ((rho - 1) u + w) e^rho - (u - rho w) e^-rho - q t = 0,
// increasing on the interval where a, b > 0 (Friberg 2021 derives the same equation; the reference's h(rho)).
// The projection onto K is a g(rho), onto the polar cone b n(rho).
//
// Behaviour that belongs to the reference's observable results and is therefore kept (not its code):
//  * two closed-form candidates per cone -- the point's own "snap" onto K (keep (u, w), lift t to the surface)
//    or onto the trivial face, likewise for the polar cone -- are returned as they are when (u, w) <= 0, when
//    one of them is within 1e-8 of v, or when the pair already is a Moreau decomposition to 1e-8
//    (exp_cone.c:395-411): points that close to the surface are NOT refined by the root search;
//  * the root's point replaces a candidate only if it is at least as close to v (exp_cone.c:421-437), and
//    rays with e^(+-rho) >= 1e15 are discarded (exp_cone.c:320-367);
//  * the power cone stops its Newton iteration on r at |f| < 1e-9 or after 20 steps (cones.c:1309-1330): the
//    returned point carries that residual, so the iteration is followed step for step -- wave-wide, lanes
//    freezing at their own stopping step.
#pragma once
#ifndef SCSAMD_EXPPOW_HOST_CHECK
#include "common.h"
#endif

namespace scsamd {

#if defined(__HIPCC__) || defined(SCSAMD_EXPPOW_HOST_CHECK)
#ifdef SCSAMD_EXPPOW_HOST_CHECK // tests/host_check_exp_pow.cpp: the same arithmetic, one lane at a time, on the host
#define XP_DEV inline
#define XP_ANY(pred) (pred)
#else
#define XP_DEV __device__ __forceinline__
#define XP_ANY(pred) (__any(pred) != 0)
#endif
typedef scs_float xreal;

struct Triple {
  xreal u, w, t;
};

namespace xp {
constexpr xreal SNAP_TOL = (xreal)1e-8;   // candidates this close are final (exp_cone.c:379)
constexpr xreal RAY_CUTOFF = (xreal)1e15; // e^(+-rho) beyond this: the ray is not used (exp_cone.c:36)
constexpr xreal FAR = (xreal)1e30;        // "no candidate" distance
constexpr int EXPAND_TRIPS = 11;          // open end of the search interval: probe at 1, 2, 4, ... 1024 from the closed end
constexpr int NEWTON_TRIPS = sizeof(xreal) == 8 ? 80 : 40; // upper bound; a wave leaves as soon as all its lanes have converged
XP_DEV xreal sel(bool c, xreal a, xreal b) { return c ? a : b; }
XP_DEV xreal vmin(xreal a, xreal b) { return a < b ? a : b; }
XP_DEV xreal vmax(xreal a, xreal b) { return a > b ? a : b; }
XP_DEV xreal vabs(xreal a) { return a < 0 ? -a : a; }
XP_DEV xreal sq(xreal a) { return a * a; }
XP_DEV xreal dist2(const Triple &a, const Triple &b) { return sq(a.u - b.u) + sq(a.w - b.w) + sq(a.t - b.t); }
XP_DEV xreal rho_cap() { return sizeof(xreal) == 8 ? (xreal)700 : (xreal)85; }

#ifdef SCSAMD_EXPPOW_HOST_CHECK
static long xp_eval_count = 0; // host check build only: evaluations of F spent by live lanes (trip histogram of the root search)
#endif
// F(rho) and F'(rho): one exponential, one reciprocal
XP_DEV void eval_F(const Triple &v, xreal rho, xreal &F, xreal &dF) {
  const xreal e = exp(rho), ei = (xreal)1 / e;
  const xreal ga = (rho - 1) * v.u + v.w; // q a
  const xreal gb = v.u - rho * v.w;       // q b
  F = ga * e - gb * ei - (rho * (rho - 1) + 1) * v.t;
  dF = (ga + v.u) * e + (gb + v.w) * ei - (2 * rho - 1) * v.t;
}

// closed-form candidate in K: the trivial face (min(u,0), 0, max(t,0)), or -- if closer -- v itself with t lifted to
// the surface w e^(u/w) (exp_cone.c:160-182)
XP_DEV xreal snap_primal(const Triple &v, Triple &p) {
  p.u = vmin(v.u, (xreal)0);
  p.w = 0;
  p.t = vmax(v.t, (xreal)0);
  xreal d = dist2(v, p);
  const bool up = v.w > (xreal)0;
  const xreal ws = sel(up, v.w, (xreal)1);
  const xreal lift = vmax(v.t, ws * exp(v.u / ws));
  const bool take = up && sq(lift - v.t) < d;
  p.u = sel(take, v.u, p.u);
  p.w = sel(take, v.w, p.w);
  p.t = sel(take, lift, p.t);
  return sel(take, sq(lift - v.t), d);
}
// closed-form candidate in the polar cone: (0, min(w,0), min(t,0)), or v with t pushed down to -u e^(w/u - 1)
// (exp_cone.c:185-207)
XP_DEV xreal snap_polar(const Triple &v, Triple &d3) {
  d3.u = 0;
  d3.w = vmin(v.w, (xreal)0);
  d3.t = vmin(v.t, (xreal)0);
  xreal d = dist2(v, d3);
  const bool up = v.u > (xreal)0;
  const xreal us = sel(up, v.u, (xreal)1);
  const xreal sink = vmin(v.t, -us * exp(v.w / us - (xreal)1));
  const bool take = up && sq(v.t - sink) < d;
  d3.u = sel(take, v.u, d3.u);
  d3.w = sel(take, v.w, d3.w);
  d3.t = sel(take, sink, d3.t);
  return sel(take, sq(v.t - sink), d);
}

// Root of F on the interval where both Moreau coefficients are positive.  `live`: this lane takes part (the others
// run along with frozen state).  The interval: a > 0 bounds rho from below (u > 0) or above (u < 0) at 1 - w/u, b > 0
// bounds it from above (w > 0) or below (w < 0) at u/w; whenever (u, w) is not <= 0 at least one end is finite, F < 0
// at a finite lower end and F > 0 at a finite upper end.  An open end is closed by probing at doubling distances.
XP_DEV xreal root_of_F(const Triple &v, bool live) {
  const xreal cap = rho_cap();
  const bool up = v.u > 0, un = v.u < 0, wp = v.w > 0, wn = v.w < 0;
  const xreal e1 = (xreal)1 - v.w / sel(up || un, v.u, (xreal)1); // end from a = 0
  const xreal e2 = v.u / sel(wp || wn, v.w, (xreal)1);            // end from b = 0
  xreal lo = vmax(sel(up, e1, -cap), sel(wn, e2, -cap));
  xreal hi = vmin(sel(un, e1, cap), sel(wp, e2, cap));
  lo = vmin(vmax(lo, -cap), cap);
  hi = vmax(vmin(hi, cap), -cap);
  const bool open_hi = !(un || wp), open_lo = !(up || wn);
  { // close the open end (at most one is open)
    const xreal anchor = sel(open_hi, lo, hi), dir = sel(open_hi, (xreal)1, (xreal)-1);
    bool grow = live && (open_hi || open_lo);
    xreal reach = 1;
    xreal near = anchor; // the bracket end on the anchor's side
    for (int k = 0; k < EXPAND_TRIPS; ++k) {
      if (!XP_ANY(grow)) break;
      const xreal probe = vmax(vmin(anchor + dir * reach, cap), -cap);
      xreal F, dF;
      eval_F(v, probe, F, dF);
#ifdef SCSAMD_EXPPOW_HOST_CHECK
      if (grow) ++xp_eval_count;
#endif
      const bool beyond = dir > 0 ? !(F < 0) : F < 0; // the probe is past the root: it closes the bracket
      if (grow) {
        if (beyond) {
          if (dir > 0) { lo = near; hi = probe; } else { hi = near; lo = probe; }
          grow = false;
        } else {
          near = probe;
          reach *= 2;
        }
      }
    }
    if (grow) { // never crossed within reach: search what is left up to the cap
      if (dir > 0) { lo = near; hi = cap; } else { hi = near; lo = -cap; }
    }
  }
  const bool proper = lo < hi; // degenerate interval: the root is the end itself
  xreal x = sel(proper, (xreal)0.5 * (lo + hi), lo), last = hi - lo;
  bool run = live && proper;
  const xreal ulp = sizeof(xreal) == 8 ? (xreal)4.5e-16 : (xreal)2.4e-7;
  for (int k = 0; k < NEWTON_TRIPS; ++k) {
    if (!XP_ANY(run)) break;
    xreal F, dF;
    eval_F(v, x, F, dF);
    const bool below = F < 0;
    const xreal nlo = sel(below, x, lo), nhi = sel(below, hi, x);
    const xreal xn = x - F / dF;
    const xreal mid = (xreal)0.5 * (nlo + nhi);
    // the Newton point is taken when it lands strictly inside the bracket and at least halves the previous move
    const bool newton_ok = dF > 0 && xn > nlo && xn < nhi && vabs(xn - x) <= (xreal)0.5 * last;
    const xreal nx = sel(newton_ok, xn, mid);
    // converged by Newton's own measure: the step has dropped below the resolution of x.  Without this exit a step that rounds
    // to x (x is then one end of the still-wide bracket, so `xn > nlo && xn < nhi` fails on the strict test) sent the lane to the
    // midpoint and it bisected ~50 more trips from there; the loop only leaves when all 64 lanes have settled, so nearly every
    // wave paid them (eval_F calls per searched triple, 20000 random triples: median 11, p90 56, p99 60 before;
    // see tests/test_exp_pow_host.py::test_root_search_trip_histogram for the figures after)
    // (fp32: an eighth of a step of resolution -- with a whole one the fp32 build's worst output error against the fp64 projection
    // grew from 7e-5 to 5e-4, the reference's fp32 build sits at 2e-4)
    const xreal tiny = (sizeof(xreal) == 8 ? (xreal)1 : (xreal)0.125) * ulp * vmax((xreal)1, vabs(x));
    const bool tiny_step = dF > 0 && (xn == x || vabs(F) <= dF * tiny);
    const bool settled = F == 0 || nx == x || tiny_step || nhi - nlo <= ulp * vmax((xreal)1, vabs(x));
#ifdef SCSAMD_EXPPOW_HOST_CHECK
    if (run) ++xp_eval_count;
#endif
    if (run) {
      lo = nlo;
      hi = nhi;
      if (!settled) {
        last = vabs(nx - x);
        x = nx;
      }
      run = !settled;
    }
  }
  return x;
}

// the points the root's rays give: in K, a g(rho); in the polar cone, b n(rho)
XP_DEV xreal ray_primal(const Triple &v, xreal rho, Triple &p) {
  const xreal e = exp(rho), ga = (rho - 1) * v.u + v.w, q = rho * (rho - 1) + 1;
  const bool ok = ga > 0 && vabs(e) < RAY_CUTOFF;
  const xreal a = ga / q;
  p.u = a * rho;
  p.w = a;
  p.t = a * e;
  return sel(ok, dist2(p, v), FAR);
}
XP_DEV xreal ray_polar(const Triple &v, xreal rho, Triple &d3) {
  const xreal ei = exp(-rho), gb = v.u - rho * v.w, q = rho * (rho - 1) + 1;
  const bool ok = gb > 0 && vabs(ei) < RAY_CUTOFF;
  const xreal b = gb / q;
  d3.u = b;
  d3.w = b * ((xreal)1 - rho);
  d3.t = -b * ei;
  return sel(ok, dist2(v, d3), FAR);
}

// `is_exp`: this lane holds an exponential-cone triple; dual = project onto the dual cone K* = -(polar cone)
XP_DEV Triple project_exp(Triple v, bool is_exp, bool dual) {
  if (dual) { v.u = -v.u; v.w = -v.w; v.t = -v.t; } // Proj_K*(v) = -Proj_polar(-v)
  Triple P, D;
  const xreal dP = snap_primal(v, P), dD = snap_polar(v, D);
  const xreal gap = vmax(vmax(vabs(P.u + D.u - v.u), vabs(P.w + D.w - v.w)), vabs(P.t + D.t - v.t));
  const bool final_snap = (v.u <= 0 && v.w <= 0) || vmin(dP, dD) <= SNAP_TOL * SNAP_TOL ||
                          (gap <= SNAP_TOL && P.u * D.u + P.w * D.w + P.t * D.t <= SNAP_TOL);
  const bool search = is_exp && !final_snap;
  if (XP_ANY(search)) {
    const xreal rho = root_of_F(v, search);
    Triple R;
    if (XP_ANY(search && !dual)) {
      const xreal dR = ray_primal(v, rho, R);
      if (search && !dual && dR <= dP) P = R;
    }
    if (XP_ANY(search && dual)) {
      const xreal dR = ray_polar(v, rho, R);
      if (search && dual && dR <= dD) D = R;
    }
  }
  Triple out;
  out.u = dual ? -D.u : P.u;
  out.w = dual ? -D.w : P.w;
  out.t = dual ? -D.t : P.t;
  return out;
}

// ---- power cone {x^a y^(1-a) >= |z|, x, y >= 0} -------------------------------------------------------------------
constexpr xreal POW_TOL = (xreal)1e-9; // membership slack and Newton stopping residual (cones.c:1291)
constexpr int POW_TRIPS = 20;
// coordinate on the optimality curve for a trial |z| = r (cones.c:1284-1288)
XP_DEV xreal pow_coord(xreal r, xreal c0, xreal r0, xreal a) {
  return vmax((xreal)0.5 * (c0 + sqrt(c0 * c0 + 4 * a * (r0 - r) * r)), (xreal)1e-12);
}
XP_DEV Triple project_pow(const Triple &v, xreal a, bool live) {
  const xreal x0 = v.u, y0 = v.w, r0 = vabs(v.t), a1 = (xreal)1 - a;
  // inside the cone / inside the polar cone: decided on clamped operands so that idle lanes never see pow(negative)
  const bool pos = x0 >= 0 && y0 >= 0, neg = x0 <= 0 && y0 <= 0;
  const xreal ax = vabs(x0), ay = vabs(y0);
  const xreal mono = pow(ax, a) * pow(ay, a1);
  const bool inside = pos && POW_TOL + mono >= r0;
  const bool in_polar = !inside && neg && POW_TOL + mono >= r0 * pow(a, a) * pow(a1, a1);
  bool run = live && !inside && !in_polar;
  xreal x = 0, y = 0, r = (xreal)0.5 * r0;
  for (int k = 0; k < POW_TRIPS; ++k) {
    if (!XP_ANY(run)) break;
    const xreal xs = pow_coord(r, x0, r0, a), ys = pow_coord(r, y0, r0, a1);
    const xreal xa = pow(xs, a), yb = pow(ys, a1);
    const xreal f = xa * yb - r;
    const xreal slope = (r0 - 2 * r);
    const xreal dx = a * slope / (2 * xs - x0), dy = a1 * slope / (2 * ys - y0);
    const xreal df = xa * yb * (a * dx / xs + a1 * dy / ys) - (xreal)1;
    const xreal rn = vmin(vmax(r - f / df, (xreal)0), r0);
    if (run) {
      x = xs;
      y = ys;
      const bool stop = vabs(f) < POW_TOL;
      r = sel(stop, r, rn);
      run = !stop;
    }
  }
  Triple out;
  out.u = inside ? v.u : (in_polar ? (xreal)0 : x);
  out.w = inside ? v.w : (in_polar ? (xreal)0 : y);
  out.t = inside ? v.t : (in_polar ? (xreal)0 : (v.t < 0 ? -r : r));
  return out;
}
} // namespace xp

#ifdef __HIPCC__
// [ep exponential | ed dual exponential | psize power (a < 0: dual power cone)], 3 consecutive rows per cone.
// A workgroup moves 256 cones per tile through LDS (unit-stride global traffic), one cone per lane.
__global__ __launch_bounds__(SCSAMD_BLOCK) void k_exp_pow(scs_float *x, int ep, int ed, int psize,
                                                          const scs_float *__restrict__ pw) {
  __shared__ scs_float tile[3 * SCSAMD_BLOCK];
  const int total = ep + ed + psize, tid = threadIdx.x;
  const int ntiles = (total + SCSAMD_BLOCK - 1) / SCSAMD_BLOCK;
  for (int tb = blockIdx.x; tb < ntiles; tb += gridDim.x) {
    const int c0 = tb * SCSAMD_BLOCK;
    const long long v0 = 3LL * c0, vend = 3LL * (c0 + SCSAMD_BLOCK < total ? c0 + SCSAMD_BLOCK : total);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const long long g = v0 + j * SCSAMD_BLOCK + tid;
      tile[j * SCSAMD_BLOCK + tid] = g < vend ? x[g] : (scs_float)0;
    }
    __syncthreads();
    const int c = c0 + tid;
    const bool have = c < total;
    Triple v{tile[3 * tid], tile[3 * tid + 1], tile[3 * tid + 2]};
    const bool is_exp = have && c < ep + ed, is_pow = have && !is_exp;
    Triple out = v;
    if (XP_ANY(is_exp)) {
      const Triple r = xp::project_exp(v, is_exp, is_exp && c >= ep);
      if (is_exp) out = r;
    }
    if (XP_ANY(is_pow)) {
      const xreal a_raw = is_pow ? pw[c - ep - ed] : (xreal)0.5;
      const bool dualp = a_raw < 0; // dual power cone: Moreau, v + Proj_K(-v) (cones.c:1427-1441)
      const xreal a = dualp ? -a_raw : a_raw;
      const Triple in{dualp ? -v.u : v.u, dualp ? -v.w : v.w, dualp ? -v.t : v.t};
      const Triple r = xp::project_pow(in, a, is_pow);
      if (is_pow) {
        out.u = dualp ? v.u + r.u : r.u;
        out.w = dualp ? v.w + r.w : r.w;
        out.t = dualp ? v.t + r.t : r.t;
      }
    }
    tile[3 * tid] = out.u;
    tile[3 * tid + 1] = out.w;
    tile[3 * tid + 2] = out.t;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const long long g = v0 + j * SCSAMD_BLOCK + tid;
      if (g < vend) x[g] = tile[j * SCSAMD_BLOCK + tid];
    }
    __syncthreads();
  }
}
#endif // __HIPCC__
#endif

} // namespace scsamd
