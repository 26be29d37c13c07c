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
// Circle method: pair i of step s+1 takes its first player from pair i+1 of step s (pair 0 keeps player 0; the last pair takes
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
PSD_HD void jacobi_cs(real d, real b, real &c, real &s) {
  const real g = d * d + b * b;
  const real lo = sizeof(real) == 8 ? (real)1e-290 : (real)1e-30, hi = sizeof(real) == 8 ? (real)1e290 : (real)1e30;
  if (g > lo && g < hi) {
    const real q = rsqrt(g);
    const real u = (real)0.5 + (real)0.5 * psd_abs(d) * q;
    const real rc = rsqrt(u);
    c = u * rc;
    s = (d >= 0 ? b : -b) * ((real)0.5 * q * rc);
  } else {
    const real h = sqrt(g);
    const real t = (d >= 0 ? b : -b) / (psd_abs(d) + h);
    c = rsqrt(t * t + (real)1);
    s = t * c;
  }
}

// The (block, row-pair) items of a lane: which 2x2 blocks (rows of pair P, columns of pair Q, P <= Q) and which (row i, pair Q) items of V
// it owns.  They do not depend on the step, and forming them takes integer divisions by run-time values: computed ONCE per
// projection, not once per step.
//
// Symmetric storage (round 5): only the entries (r, c) with r <= c of A are kept current -- entry (r, c) lives at min(r, c) * ld +
// max(r, c) -- so a step updates npairs (npairs + 1) / 2 blocks instead of npairs^2, with half the LDS traffic and half the
// arithmetic on A, and A stays symmetric by construction (rounds 2-4 computed both triangles, equal only to rounding, and read one).
// The lower triangle is never read after the warm start.
template <int NB>
struct PsdItems {
  int bP[NB], bQ[NB], vQ[3 * NB], vI[3 * NB];
  bool okb[NB], okv[3 * NB];
};
// blocks per lane for `nthreads` update lanes: the upper triangle of the npairs x npairs block grid; the V items (2 npairs^2 of them)
// need up to 3 per block slot (npairs^2 * 2 <= 3 * NB * nthreads whenever npairs (npairs + 1) / 2 <= NB * nthreads and npairs >= 3 ...
// checked by psd_items_fit)
PSD_HD int psd_blocks_per_lane(int npairs, int nthreads) {
  int nb = (npairs * (npairs + 1) / 2 + nthreads - 1) / nthreads;
  while (3 * nb * nthreads < 2 * npairs * npairs) ++nb;
  return nb;
}
template <int NB>
PSD_HD void psd_items_init(PsdItems<NB> &it, int tid, int nthreads, int npairs, int K2) {
  const int ntri = npairs * (npairs + 1) / 2, nv = 2 * npairs * npairs; // K2 * npairs == 2 * npairs^2
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
  PSD_UNROLL
  for (int j = 0; j < 3 * NB; ++j) {
    const int f = tid + j * nthreads;
    it.okv[j] = tid < nthreads && f < nv;
    const int fc = it.okv[j] ? f : 0;
    it.vQ[j] = fc / K2;
    it.vI[j] = fc % K2; // consecutive rows: stride ld
  }
}

// where entry (r, c) of the symmetric matrix is stored
PSD_HD int psd_sym_index(int r, int c, int ld) { return r < c ? r * ld + c : c * ld + r; }

// One update pass, A_dst <- J' A_src J over the 2x2 blocks (rows of pair P, columns of pair Q, P <= Q) and V <- V J over (row, pair)
// items: every lane owns up to NB blocks and 3 NB row pairs and asks for all its tables, then all its operands, before it computes
// -- the LDS round trips of a lane's items overlap instead of queueing behind each other.  Items beyond the end are clamped to item
// 0 for the loads and skipped by the stores (no per-item branches in the load phase).  A_dst == A_src is the in-place form (every
// entry is read and written by the same lane); the pipelined step passes the other copy.
template <int NB>
PSD_HD void psd_update_pass(const real *Asrc, real *Adst, real *V, const PsdPair *rot_pq, const RotCS *rot_cs, const PsdItems<NB> &it,
                            int ld, bool rotates = true) {
  constexpr int NV = 3 * NB;
  int i11[NB], i12[NB], i21[NB], i22[NB];
  RotCS r1[NB], r2[NB];
  bool diag[NB];
  PSD_UNROLL
  for (int u = 0; u < NB; ++u) {
    const PsdPair pq1 = rot_pq[it.bP[u]], pq2 = rot_pq[it.bQ[u]];
    r1[u] = rot_cs[it.bP[u]];
    r2[u] = rot_cs[it.bQ[u]];
    diag[u] = it.bP[u] == it.bQ[u];
    i11[u] = psd_sym_index(pq1.x, pq2.x, ld);
    i12[u] = psd_sym_index(pq1.x, pq2.y, ld);
    i21[u] = psd_sym_index(pq1.y, pq2.x, ld); // diagonal block: the same stored entry as i12 (x < y)
    i22[u] = psd_sym_index(pq1.y, pq2.y, ld);
  }
  int ip[NV], iq[NV];
  RotCS rq[NV];
  PSD_UNROLL
  for (int j = 0; j < NV; ++j) {
    const PsdPair pq2 = rot_pq[it.vQ[j]];
    rq[j] = rot_cs[it.vQ[j]];
    ip[j] = it.vI[j] * ld + pq2.x;
    iq[j] = it.vI[j] * ld + pq2.y;
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
  // (pipelined step: `rotates` comes out of LDS too -- asked for before the tables, needed only here, so that its round trip rides
  // along with the loads instead of preceding them; a step that rotates nothing stores nothing)
  if (!rotates) return;
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
  }
  PSD_UNROLL
  for (int j = 0; j < NV; ++j) {
    if (it.okv[j]) {
      V[ip[j]] = rq[j].c * vp[j] - rq[j].s * vq[j];
      V[iq[j]] = rq[j].s * vp[j] + rq[j].c * vq[j];
    }
  }
}

// the players of pair i at the round-robin positions (pos_a, pos_b), and the positions of the next step (player 0 never moves;
// the others advance by one around the K2 - 1 seats)
PSD_HD void psd_pair_advance(int i, int K2, int &pos_a, int &pos_b) {
  if (i != 0) pos_a = pos_a == K2 - 1 ? 1 : pos_a + 1;
  pos_b = pos_b == K2 - 1 ? 1 : pos_b + 1;
}

// rotation of the pair (lo, hi), lo < hi, from its three entries; entries at or below the threshold -- and the padding index of an
// odd order -- are left alone (identity).  Returns whether it rotates; tracks the sweep's largest off-diagonal entry.
PSD_HD bool psd_make_rotation(real apq, real app, real aqq, int hi, int k, real thr, real &offmax, RotCS &cs) {
  real c = 1, s = 0;
  const real aa = psd_abs(apq);
  bool rot = false;
  if (hi < k) {
    offmax = aa > offmax ? aa : offmax;
    if (aa > thr) {
      jacobi_cs(aqq - app, (real)2 * apq, c, s);
      rot = true;
    }
  }
  cs = RotCS{c, s};
  return rot;
}

// one entry (row side rs, column side cs_) of the 2x2 block J1' [a11 a12; a21 a22] J2 -- the expressions of psd_update_pass
PSD_HD real psd_block_entry(real a11, real a12, real a21, real a22, const RotCS &r1, const RotCS &r2, int rs, int cs_) {
  const real ra = rs == 0 ? r1.c * a11 - r1.s * a21 : r1.s * a11 + r1.c * a21; // row rs of J1' A, column 0
  const real rb = rs == 0 ? r1.c * a12 - r1.s * a22 : r1.s * a12 + r1.c * a22; // column 1
  return cs_ == 0 ? r2.c * ra - r2.s * rb : r2.s * ra + r2.c * rb;
}

// one pair's record: its players (x < y) and its rotation
struct PsdRot {
  int x, y;
  real c, s;
};

// Look-ahead of lane i (pair i of step s + 1; (p, q) = its players, any order) from the records of the two pairs of step s that hold p
// and q: reads the matrix as it stands BEFORE step s is applied, returns the rotation of its pair for the matrix AFTER step s.
// p_first: the pair holding p has the smaller pair index (only for lane 0).  In the kernel the records come out of the look-ahead
// wave's OWN registers -- lane i wrote pair i of step s one phase earlier, its neighbours' records arrive by DPP lane shifts -- so the
// chain of a step holds ONE LDS round trip (the ten matrix entries), issued at the top of the phase before the update waves' traffic
// (round 5: with the records read back from the LDS tables the look-ahead wave was the critical path, 1 900 of a step's 2 450 clocks,
// its two dependent round trips queueing behind seven waves' worth of update traffic; profiles/r5_psd_pipelined_step.md).
PSD_HD bool psd_lookahead_rec(const real *A, const PsdRot &rec_p, const PsdRot &rec_q, bool p_first, int p, int q, int ld, int k, real thr,
                              real &offmax, PsdPair &pq_out, RotCS &cs_out) {
  const bool swap = p > q;
  const int lo = swap ? q : p, hi = swap ? p : q;
  const PsdRot &pl = swap ? rec_q : rec_p, &ph = swap ? rec_p : rec_q; // the pairs of step s that hold lo / hi
  const bool lo_first = swap ? !p_first : p_first;                      // pair index of lo's pair <= that of hi's pair
  const RotCS rl{pl.c, pl.s}, rh{ph.c, ph.s};
  const int sl = lo == pl.x ? 0 : 1, sh = hi == ph.x ? 0 : 1; // which player of its old pair
  // the three 2x2 blocks that hold a_lohi, a_lolo, a_hihi: (Plo, Phi), (Plo, Plo), (Phi, Phi) -- ten independent reads
  // (symmetric storage: entry (r, c) lives at min * ld + max; a pair's own block holds its off-diagonal entry once)
  const real b11 = A[psd_sym_index(pl.x, ph.x, ld)], b12 = A[psd_sym_index(pl.x, ph.y, ld)];
  const real b21 = A[psd_sym_index(pl.y, ph.x, ld)], b22 = A[psd_sym_index(pl.y, ph.y, ld)];
  const real l11 = A[pl.x * ld + pl.x], l12 = A[pl.x * ld + pl.y], l21 = l12, l22 = A[pl.y * ld + pl.y];
  const real h11 = A[ph.x * ld + ph.x], h12 = A[ph.x * ld + ph.y], h21 = h12, h22 = A[ph.y * ld + ph.y];
  // the update forms the block with the rows of the pair of SMALLER pair index: the same orientation here, so that the three entries
  // carry exactly the bits the update stores (the pipelined iteration then equals the two-phase one bit for bit)
  const real apq = lo_first ? psd_block_entry(b11, b12, b21, b22, rl, rh, sl, sh) : psd_block_entry(b11, b21, b12, b22, rh, rl, sh, sl);
  const real app = psd_block_entry(l11, l12, l21, l22, rl, rl, sl, sl);
  const real aqq = psd_block_entry(h11, h12, h21, h22, rh, rh, sh, sh);
  pq_out.x = lo;
  pq_out.y = hi;
  return psd_make_rotation(apq, app, aqq, hi, k, thr, offmax, cs_out);
}
// which pairs of step s hold the players of pair i of step s + 1 (circle method, see the header)
PSD_HD int psd_source_of_p(int i, int npairs) { return i == 0 ? 0 : (i == npairs - 1 ? npairs - 1 : i + 1); }
PSD_HD int psd_source_of_q(int i) { return i == 0 ? 1 : i - 1; }

// first step of a sweep: nothing is pending, the three entries are read as they stand
PSD_HD bool psd_first_rotation(const real *A, int p, int q, int ld, int k, real thr, real &offmax, PsdPair &pq_out, RotCS &cs_out) {
  const int lo = p > q ? q : p, hi = p > q ? p : q;
  pq_out.x = lo;
  pq_out.y = hi;
  const real apq = A[lo * ld + hi];
  real app = 0, aqq = 0;
  if (hi < k && psd_abs(apq) > thr) {
    app = A[lo * ld + lo];
    aqq = A[hi * ld + hi];
  }
  return psd_make_rotation(apq, app, aqq, hi, k, thr, offmax, cs_out);
}

} // namespace scsamd
