// psd_lds_step.h -- the lane-level pieces of ONE round-robin step of the LDS Jacobi kernel (k_psd_jacobi, cones.hip):
// rotation parameters, the 2x2-block update pass, and (round 5) the LOOK-AHEAD that forms step s+1's rotations while step s
// is still being applied.  Replaces the eigen-decomposition behind reference src/cones.c:999-1067 (LAPACK dsyevr).
//
// The same functions compile for the host (PSD_STEP_HOST_CHECK, tests/native/host_check_psd_step.cpp: lanes looped one at a
// time, a barrier = the end of a loop), so the schedule -- which pair feeds which, the ping-pong of A, the table parities --
// is pinned on the CPU against numpy's eigh and against the un-pipelined order without a GPU.
//
// Pipelined step (orders K2 <= PSD_WARM_KMAX, where a second copy of A fits LDS beside A and V):
//   tables[t & 1]   = (p, q, c, s) of every pair of step t;  rot_any[t & 1] = does step t rotate at all
//   phase s (ONE barrier per step instead of two):
//     update lanes    read  A[cur], tables[s & 1];  write A[cur ^ 1] (all of it) and V in place      -- only if rot_any[s & 1]
//     look-ahead lanes (their own wave) read A[cur], tables[s & 1]; apply step s's rotations to just the three entries
//                     a_pq, a_pp, a_qq their pair of step s+1 needs; write tables[(s+1) & 1], rot_any[(s+1) & 1]
//   barrier;  cur ^= rot_any[s & 1]
// (Pairs are kept in position order (x, y), not sorted.)  Circle method: pair i of step s+1 takes its first player from pair i+1 of step s (pair 0 keeps player 0; the last pair takes
// the OTHER player of itself) and its second player from pair i-1 (pair 0: from pair 1) -- so the look-ahead lane knows which two
// table rows to read without an inverse table.
#pragma once

#ifdef PSD_STEP_HOST_CHECK
#define PSD_HD inline
#define PSD_UNROLL
struct PsdPair {
  int x, y;
};
#else
#define PSD_HD __device__ __forceinline__
#define PSD_UNROLL _Pragma("unroll")
typedef int2 PsdPair;
#endif

namespace scsamd {

struct alignas(2 * sizeof(real)) RotCS {
  real c, s;
};

PSD_HD real psd_abs(real x) { return x < 0 ? -x : x; }

// (c, s) of the Jacobi rotation that annihilates a_pq: t = sgn(d) b / (|d| + h), h = sqrt(d^2 + b^2), d = a_qq - a_pp, b = 2 a_pq,
// c = 1 / sqrt(1 + t^2), s = t c.  Written through the identities (|d| + h)^2 + b^2 = 2 h (|d| + h) and c^2 = (h + |d|) / (2 h):
//     q = 1 / h = rsqrt(d^2 + b^2),  u = 1/2 + |d| q / 2 = c^2,  rc = rsqrt(u),  c = u rc,  s = sgn(d) b q rc / 2
// -- two reciprocal square roots instead of a square root, a division and a reciprocal square root in sequence: on this part each of
// them is a Newton sequence, and the chain sits on the critical path of every Jacobi step.  c^2 + s^2 = 1 holds to
// rounding as before.  Outside the range where d^2 + b^2 is a normal number the sequential form is kept.
PSD_HD void jacobi_cs(real d, real b, real &c, real &s, real &t) { // t = s / c = tan of the rotation angle
  const real g = d * d + b * b;
  const real lo = sizeof(real) == 8 ? (real)1e-290 : (real)1e-30, hi = sizeof(real) == 8 ? (real)1e290 : (real)1e30;
  if (g > lo && g < hi) {
    const real q = rsqrt(g);
    const real u = (real)0.5 + (real)0.5 * psd_abs(d) * q;
    const real rc = rsqrt(u); // = 1 / c
    c = u * rc;
    s = (d >= 0 ? b : -b) * ((real)0.5 * q * rc);
    t = s * rc;
  } else {
    const real h = sqrt(g);
    t = (d >= 0 ? b : -b) / (psd_abs(d) + h);
    c = rsqrt(t * t + (real)1);
    s = t * c;
  }
}
PSD_HD void jacobi_cs(real d, real b, real &c, real &s) {
  real t;
  jacobi_cs(d, b, c, s, t);
}

// The (block, row-pair) items of a lane: which 2x2 blocks (rows of pair P, columns of pair Q, P <= Q) and which (row i, pair Q) items of V
// it owns.  They do not depend on the step, and forming them takes integer divisions by run-time values: computed ONCE per
// projection, not once per step.
//
// Symmetric storage (round 5): only the entries (r, c) with r <= c of A are kept current -- entry (r, c) lives at min(r, c) * ld +
// max(r, c) -- so a step updates npairs (npairs + 1) / 2 blocks instead of npairs^2, with half the LDS traffic and half the
// arithmetic on A, and A stays symmetric by construction (rounds 2-4 computed both triangles, equal only to rounding, and read one).
// The lower triangle is never read after the warm start.
template <int NB, int VR = 3> // VR: rows of V per lane and block (3 ships; 4 with six update waves, the look-ahead wave alone on its SIMD)
struct PsdItems {
  int bP[NB], bQ[NB], vQ, vI[VR * NB];
  bool okb[NB], okv[VR * NB];
};
// blocks per lane for `nthreads` update lanes: the upper triangle of the npairs x npairs block grid, NB blocks per lane; the (row,
// pair) items of V: a lane takes up to 3 NB consecutive rows of ONE pair Q (one table read serves them all), so npairs *
// ceil(K2 / (3 NB)) lanes must exist
PSD_HD int psd_v_groups(int K2, int nb, int vr = 3) { return (K2 + vr * nb - 1) / (vr * nb); }
PSD_HD int psd_blocks_per_lane(int npairs, int nthreads, int vr = 3) {
  int nb = (npairs * (npairs + 1) / 2 + nthreads - 1) / nthreads;
  while (npairs * psd_v_groups(2 * npairs, nb, vr) > nthreads) ++nb;
  return nb;
}
template <int NB, int VR>
PSD_HD void psd_items_init(PsdItems<NB, VR> &it, int tid, int nthreads, int npairs, int K2) {
  const int ntri = npairs * (npairs + 1) / 2;
  PSD_UNROLL
  for (int u = 0; u < NB; ++u) {
    const int e = tid + u * nthreads;
    it.okb[u] = tid < nthreads && e < ntri;
    // e -> (P, Q), P <= Q, column-major over the upper triangle: Q = the column with Q (Q + 1) / 2 <= e, P = e - Q (Q + 1) / 2
    // (consecutive lanes walk the ROW pairs of one column pair: distinct rows, stride ld (odd) -> distinct LDS banks)
    int Q = 0;
    if (it.okb[u]) {
      while ((Q + 1) * (Q + 2) / 2 <= e) ++Q;
    }
    it.bQ[u] = Q;
    it.bP[u] = it.okb[u] ? e - Q * (Q + 1) / 2 : 0;
  }
  const int G = psd_v_groups(K2, NB, VR); // lanes per pair Q
  const bool lane_ok = tid < nthreads && tid < npairs * G;
  const int tq = lane_ok ? tid : 0;
  it.vQ = tq / G;
  const int i0 = (tq % G) * VR * NB;
  PSD_UNROLL
  for (int j = 0; j < VR * NB; ++j) {
    it.okv[j] = lane_ok && i0 + j < K2;
    it.vI[j] = it.okv[j] ? i0 + j : 0; // consecutive lanes: rows 3 NB apart, stride 3 NB ld
  }
}

// ---- round 6, "signal" form of the pipelined step (k_psd_jacobi<2>) ---------------------------------------------------------------
// The look-ahead of round 5 RE-DERIVES, from the matrix before step s and step s's rotations, the three entries (a_pq, a_pp, a_qq) its
// pair of step s + 1 needs -- ~230 instructions of one wave, 1 550 clocks, the bound of the step.  But those entries are among the
// first things the update itself can produce: pair i of step s + 1 takes its players from pairs i + 1 / i - 1 of step s, so they live
// in the block (i - 1, i + 1) of step s's block grid ((0, 1) for pair 0, (npairs - 2, npairs - 1) for the last pair) and in the diagonal
// blocks -- 2 npairs PRIORITY blocks of npairs (npairs + 1) / 2, the same ones in every step because they are named by pair POSITION.
// In the signal form the update lanes own the priority blocks FIRST (items 0 .. npri - 1: wave 0, and wave 1 from order 66 on), store
// them, and raise a counter in LDS; the rotation wave waits for the counter and reads its three entries from the updated copy -- the
// plain rotation of psd_first_rotation, exactly what the two-phase step computes (the signal form is bit-identical to it).
PSD_HD int psd_priority_count(int npairs) { return npairs >= 3 ? 2 * npairs : (npairs == 2 ? 3 : 1); }
// item e -> block (P, Q), P <= Q: priority blocks first -- the diagonal (j, j), then (0, 1), (i - 1, i + 1) for i = 1 .. npairs - 2,
// (npairs - 2, npairs - 1) -- then the others: (P, P + 1) for P = 1 .. npairs - 3, then by distance d = Q - P = 3, 4, ...
PSD_HD void psd_block_of_item(int e, int npairs, int &P, int &Q) {
  if (e < npairs) {
    P = Q = e;
    return;
  }
  if (npairs == 2) { // (0,0), (1,1), (0,1)
    P = 0;
    Q = 1;
    return;
  }
  e -= npairs;
  if (e == 0) {
    P = 0;
    Q = 1;
    return;
  }
  if (e <= npairs - 2) { // i = e: (i - 1, i + 1)
    P = e - 1;
    Q = e + 1;
    return;
  }
  if (e == npairs - 1) {
    P = npairs - 2;
    Q = npairs - 1;
    return;
  }
  e -= npairs;
  if (e < npairs - 3) { // neighbours (P, P + 1), P = 1 .. npairs - 3
    P = e + 1;
    Q = e + 2;
    return;
  }
  e -= npairs - 3 > 0 ? npairs - 3 : 0;
  int d = 3;
  while (d < npairs && e >= npairs - d) {
    e -= npairs - d;
    ++d;
  }
  P = e;
  Q = e + d;
}
template <int NB, int VR>
PSD_HD void psd_items_init_priority(PsdItems<NB, VR> &it, int tid, int nthreads, int npairs, int K2) {
  psd_items_init<NB, VR>(it, tid, nthreads, npairs, K2); // the (row, pair) items of V, and okb
  PSD_UNROLL
  for (int u = 0; u < NB; ++u) {
    int P = 0, Q = 0;
    if (it.okb[u]) psd_block_of_item(tid + u * nthreads, npairs, P, Q);
    it.bP[u] = P;
    it.bQ[u] = Q;
  }
}
// where entry (r, c) of the symmetric matrix is stored.  (Device: the 24-bit multiply-add -- orders and leading dimensions are < 100 --
// is a full-rate instruction; the 32-bit r * ld + c compiles to v_mad_u64_u32, a quarter-rate one, four times per block of the update
// pass and six times on the look-ahead wave's chain in every step.)
#ifdef PSD_STEP_HOST_CHECK
PSD_HD int psd_mad24(int a, int b, int c) { return a * b + c; }
#else
PSD_HD int psd_mad24(int a, int b, int c) { return __mul24(a, b) + c; }
#endif
PSD_HD int psd_sym_index(int r, int c, int ld) {
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  return psd_mad24(lo, ld, hi);
}

// One update pass, A_dst <- J' A_src J over the 2x2 blocks (rows of pair P, columns of pair Q, P <= Q) and V <- V J over (row, pair)
// items: every lane owns up to NB blocks and 3 NB row pairs and asks for all its tables, then all its operands, before it computes
// -- the LDS round trips of a lane's items overlap instead of queueing behind each other.  Items beyond the end are clamped to item
// 0 for the loads and skipped by the stores (no per-item branches in the load phase).  A_dst == A_src is the in-place form (every
// entry is read and written by the same lane); the pipelined step passes the other copy.
// Returns the flag (1 without one).
template <int NB, int VR, bool SIG = false>
PSD_HD int psd_update_pass(const real *Asrc, real *Adst, real *V, const PsdPair *rot_pq, const RotCS *rot_cs, const PsdItems<NB, VR> &it,
                           int ld, const int *rotates_flag = nullptr, int *signal = nullptr) {
  // signal (signal form of the pipelined step, device only): non-null on lane 0 of the waves that own priority blocks -- raised by one
  // right after the wave's FIRST block (item u = 0: all priority blocks are items u = 0) has been stored.  A wave's LDS instructions
  // execute in program order, so whoever sees the counter raised sees the stores; the empty asm statements keep the compiler from
  // moving the stores across the atomic.
  // pipelined step: "does this step rotate at all" is a word in LDS; it is asked for together with the tables (ONE round trip for the
  // flag, the block's two pairs and the row pairs' pair -- left alone the compiler reads the flag, waits, branches, reads the block's
  // tables, waits, reads the rows' table, waits: four dependent round trips per step where two are needed)
  int flag = rotates_flag ? *rotates_flag : 1;
  constexpr int NV = VR * NB;
  int i11[NB], i12[NB], i21[NB], i22[NB];
  RotCS r1[NB], r2[NB];
  PsdPair q1[NB], q2[NB];
  bool diag[NB];
  PSD_UNROLL
  for (int u = 0; u < NB; ++u) {
    q1[u] = rot_pq[it.bP[u]];
    q2[u] = rot_pq[it.bQ[u]];
    r1[u] = rot_cs[it.bP[u]];
    r2[u] = rot_cs[it.bQ[u]];
  }
  PsdPair pqv = rot_pq[it.vQ];
  RotCS rq = rot_cs[it.vQ];
#ifndef PSD_STEP_HOST_CHECK
  // the flag and every table entry: one batch of LDS reads, one wait
  asm volatile("" : "+v"(flag), "+v"(pqv.x), "+v"(pqv.y), "+v"(rq.c), "+v"(rq.s));
  PSD_UNROLL
  for (int u = 0; u < NB; ++u)
    asm volatile("" : "+v"(q1[u].x), "+v"(q1[u].y), "+v"(q2[u].x), "+v"(q2[u].y), "+v"(r1[u].c), "+v"(r1[u].s), "+v"(r2[u].c), "+v"(r2[u].s));
#endif
  PSD_UNROLL
  for (int u = 0; u < NB; ++u) {
    diag[u] = it.bP[u] == it.bQ[u];
    i11[u] = psd_sym_index(q1[u].x, q2[u].x, ld);
    i12[u] = psd_sym_index(q1[u].x, q2[u].y, ld);
    i21[u] = psd_sym_index(q1[u].y, q2[u].x, ld); // diagonal block: the same stored entry as i12
    i22[u] = psd_sym_index(q1[u].y, q2[u].y, ld);
  }
  int ip[NV], iq[NV];
  PSD_UNROLL
  for (int j = 0; j < NV; ++j) {
    ip[j] = psd_mad24(it.vI[j], ld, pqv.x);
    iq[j] = psd_mad24(it.vI[j], ld, pqv.y);
  }
  real a11[NB], a12[NB], a21[NB], a22[NB], vp[NV], vq[NV];
  PSD_UNROLL
  for (int u = 0; u < NB; ++u) {
    a11[u] = Asrc[i11[u]];
    a12[u] = Asrc[i12[u]];
    a21[u] = Asrc[i21[u]];
    a22[u] = Asrc[i22[u]];
  }
  PSD_UNROLL
  for (int j = 0; j < NV; ++j) {
    vp[j] = V[ip[j]];
    vq[j] = V[iq[j]];
  }
  // (a step that rotates nothing stores nothing)
  if (!flag) return 0;
  PSD_UNROLL
  for (int u = 0; u < NB; ++u) {
    if (it.okb[u]) {
      const real c1 = r1[u].c, s1 = r1[u].s, c2 = r2[u].c, s2 = r2[u].s;
      const real r11 = c1 * a11[u] - s1 * a21[u], r12 = c1 * a12[u] - s1 * a22[u];
      const real r21 = s1 * a11[u] + c1 * a21[u], r22 = s1 * a12[u] + c1 * a22[u];
      Adst[i11[u]] = c2 * r11 - s2 * r12;
      Adst[i22[u]] = s2 * r21 + c2 * r22;
      if (diag[u]) {
        // the rotated pair's own off-diagonal entry is zero by construction: store the exact zero (what is left otherwise is
        // rounding residue of the order eps |a_pp - a_qq|, which for k >~ 100 sits above the convergence threshold and kept
        // the sweeps going to the cap); a pair that did not rotate keeps its entry (identity)
        Adst[i12[u]] = s1 != (real)0 ? (real)0 : s2 * r11 + c2 * r12;
      } else {
        Adst[i12[u]] = s2 * r11 + c2 * r12;
        Adst[i21[u]] = c2 * r21 - s2 * r22;
      }
    }
#ifndef PSD_STEP_HOST_CHECK
    if (SIG && u == 0) {
      asm volatile("" ::: "memory");
      if (signal) __hip_atomic_fetch_add(signal, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
    }
#endif
  }
  PSD_UNROLL
  for (int j = 0; j < NV; ++j) {
    if (it.okv[j]) {
      V[ip[j]] = rq.c * vp[j] - rq.s * vq[j];
      V[iq[j]] = rq.s * vp[j] + rq.c * vq[j];
    }
  }
  return flag;
}

// the players of pair i at the round-robin positions (pos_a, pos_b), and the positions of the next step (player 0 never moves;
// the others advance by one around the K2 - 1 seats)
PSD_HD void psd_pair_advance(int i, int K2, int &pos_a, int &pos_b) {
  if (i != 0) pos_a = pos_a == K2 - 1 ? 1 : pos_a + 1;
  pos_b = pos_b == K2 - 1 ? 1 : pos_b + 1;
}

// one pair's record as the look-ahead lanes keep it in registers: its players (x, y) in position order -- NOT sorted: the rotation
// is formed for the order (x, y) and applied in that order, any order describes the same plane rotation -- its rotation and
// t = s / c
struct PsdRot {
  int x, y;
  real c, s, t;
};

// rotation of the pair (p, q) from its three entries (a_pq, a_pp, a_qq); entries at or below the threshold -- and the padding index of
// an odd order -- are left alone (identity).  Returns whether it rotates; tracks the sweep's largest off-diagonal entry.
PSD_HD bool psd_make_rotation(real apq, real app, real aqq, int p, int q, int k, real thr, real &offmax, PsdRot &r) {
  real c = 1, s = 0, t = 0;
  const real aa = psd_abs(apq);
  bool rot = false;
  if ((p > q ? p : q) < k) {
    offmax = aa > offmax ? aa : offmax;
    if (aa > thr) {
      jacobi_cs(aqq - app, (real)2 * apq, c, s, t);
      rot = true;
    }
  }
  r = PsdRot{p, q, c, s, t};
  return rot;
}

// Look-ahead of lane i (pair i of step s + 1, players p and q) from the records of the two pairs of step s that hold p and q: reads
// the matrix as it stands BEFORE step s is applied, returns the rotation of (p, q) for the matrix AFTER step s.  In the kernel the
// records come out of the look-ahead wave's OWN registers -- lane i formed pair i of step s one phase earlier, its neighbours' records
// arrive by DPP lane shifts -- so the chain of a step holds ONE LDS round trip (eight matrix entries).
//   a'_pq = (column sp of J_P)' B (column sq of J_Q),  B = A[pair P, pair Q]                       -- six multiply-adds
//   a'_pp = a_pp -/+ t_P a_xy(P)   (p is the first / second player of P; the rotation annihilates a_xy: the classical update)
// (Round 5, measured: the look-ahead wave is bound by INSTRUCTION ISSUE -- one wave, 4 clocks per instruction -- not by LDS latency: the
// first form, which restated the update's two-level block product for all three entries and sorted the pair, took 1 900 clocks of a
// 2 450-clock step in ~430 instructions; profiles/r5_psd_pipelined_step.md.  The diagonal entries therefore differ from what the update
// stores by rounding, O(eps |a|): the rotation angle moves by that much, nothing else.)
// The look-ahead in two parts: PREPARE (no matrix entry needed: which eight entries, which coefficients -- runs at the END of the previous
// phase, so that the eight reads are the very first LDS requests of the next phase, ahead of the update waves' traffic in the CU's LDS
// queue) and FINISH (the reads, sixteen multiply-adds, the rotation).
struct PsdLaPlan {
  int a[8];                 // b11, b12, b21, b22, a_pp, a_qq, a_xy(P), a_xy(Q)
  real u0, u1, v0, v1, tp, tq;
  int p, q;
};
PSD_HD void psd_la_prepare(PsdLaPlan &pl, const PsdRot &P, const PsdRot &Q, int p, int q, int ld) {
  const bool sp = p != P.x, sq = q != Q.x; // p / q is the second player of its old pair
  pl.a[0] = psd_sym_index(P.x, Q.x, ld);
  pl.a[1] = psd_sym_index(P.x, Q.y, ld);
  pl.a[2] = psd_sym_index(P.y, Q.x, ld);
  pl.a[3] = psd_sym_index(P.y, Q.y, ld);
  pl.a[4] = psd_mad24(p, ld, p);
  pl.a[5] = psd_mad24(q, ld, q);
  pl.a[6] = psd_sym_index(P.x, P.y, ld);
  pl.a[7] = psd_sym_index(Q.x, Q.y, ld);
  // row-rotation coefficients of the update: first player (c, -s), second player (s, c)
  pl.u0 = sp ? P.s : P.c;
  pl.u1 = sp ? P.c : -P.s;
  pl.v0 = sq ? Q.s : Q.c;
  pl.v1 = sq ? Q.c : -Q.s;
  pl.tp = sp ? P.t : -P.t;
  pl.tq = sq ? Q.t : -Q.t;
  pl.p = p;
  pl.q = q;
}
PSD_HD bool psd_la_finish(const real *A, const PsdLaPlan &pl, int k, real thr, real &offmax, PsdRot &out) {
  const real b11 = A[pl.a[0]], b12 = A[pl.a[1]], b21 = A[pl.a[2]], b22 = A[pl.a[3]];
  real app0 = A[pl.a[4]], aqq0 = A[pl.a[5]], pxy = A[pl.a[6]], qxy = A[pl.a[7]];
#ifndef PSD_STEP_HOST_CHECK
  // all eight reads are issued together: left alone the compiler sinks the four diagonal reads into the "rotates" branch of
  // psd_make_rotation -- a second LDS round trip behind the first on the chain that bounds the step
  asm volatile("" : "+v"(app0), "+v"(aqq0), "+v"(pxy), "+v"(qxy));
#endif
  const real ra = pl.u0 * b11 + pl.u1 * b21, rb = pl.u0 * b12 + pl.u1 * b22;
  const real apq = pl.v0 * ra + pl.v1 * rb;
  const real app = app0 + pl.tp * pxy;
  const real aqq = aqq0 + pl.tq * qxy;
  return psd_make_rotation(apq, app, aqq, pl.p, pl.q, k, thr, offmax, out);
}
PSD_HD bool psd_lookahead_rec(const real *A, const PsdRot &P, const PsdRot &Q, int p, int q, int ld, int k, real thr, real &offmax,
                              PsdRot &out) {
  PsdLaPlan pl;
  psd_la_prepare(pl, P, Q, p, q, ld);
  return psd_la_finish(A, pl, k, thr, offmax, out);
}
// which pairs of step s hold the players of pair i of step s + 1 (circle method, see the header)
PSD_HD int psd_source_of_p(int i, int npairs) { return i == 0 ? 0 : (i == npairs - 1 ? npairs - 1 : i + 1); }
PSD_HD int psd_source_of_q(int i) { return i == 0 ? 1 : i - 1; }

// first step of a sweep (and every step of the two-phase form): nothing is pending, the three entries are read as they stand
PSD_HD bool psd_first_rotation(const real *A, int p, int q, int ld, int k, real thr, real &offmax, PsdRot &out) {
  const real apq = A[psd_sym_index(p, q, ld)];
  real app = 0, aqq = 0;
  if ((p > q ? p : q) < k && psd_abs(apq) > thr) {
    app = A[psd_mad24(p, ld, p)];
    aqq = A[psd_mad24(q, ld, q)];
  }
  return psd_make_rotation(apq, app, aqq, p, q, k, thr, offmax, out);
}

// the rotation of pair (p, q) from the matrix as it stands, its three entries asked for in ONE batch of LDS reads
PSD_HD bool psd_rotation_now(const real *A, int p, int q, int ld, int k, real thr, real &offmax, PsdRot &out) {
  real apq = A[psd_sym_index(p, q, ld)], app = A[psd_mad24(p, ld, p)], aqq = A[psd_mad24(q, ld, q)];
#ifndef PSD_STEP_HOST_CHECK
  asm volatile("" : "+v"(apq), "+v"(app), "+v"(aqq));
#endif
  return psd_make_rotation(apq, app, aqq, p, q, k, thr, offmax, out);
}


} // namespace scsamd
