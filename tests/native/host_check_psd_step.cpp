// CPU build of the LDS Jacobi step of scs_amd/csrc/psd_lds_step.h (PSD_STEP_HOST_CHECK: the same lane-level functions, lanes looped one
// at a time, a workgroup barrier = the end of a loop) so that the `-m "not gpu"` suite can pin the PIPELINED schedule of k_psd_jacobi
// (round 5: look-ahead wave, ping-pong A, one barrier per step) against the un-pipelined order of rounds 2-4 and against numpy's eigh
// without a GPU.  Test infrastructure only.
#include <cmath>
#include <cstring>
#include <vector>
typedef double scs_float;
typedef scs_float real;
#define PSD_STEP_HOST_CHECK 1
using std::sqrt;
static inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
#include "../../scs_amd/csrc/psd_lds_step.h"
using namespace scsamd;

namespace {
constexpr int TBL = 256;      // second table buffer offset, as in cones.hip (PSD_TBL)
constexpr int NT_ALL = 512;   // PSD_THREADS
constexpr int NT_PIPE = 448;  // PSD_THREADS - the look-ahead wave

template <int NB>
void update_all(const real *As, real *Ad, real *V, const PsdPair *pq, const RotCS *cs, int npairs, int K2, int ld, int nt, bool priority) {
  for (int tid = 0; tid < nt; ++tid) {
    PsdItems<NB, 3> it;
    if (priority) psd_items_init_priority<NB, 3>(it, tid, nt, npairs, K2); // the signal form's enumeration: priority blocks first
    else psd_items_init<NB, 3>(it, tid, nt, npairs, K2);
    psd_update_pass<NB, 3>(As, Ad, V, pq, cs, it, ld);
  }
}
void update_dispatch(const real *As, real *Ad, real *V, const PsdPair *pq, const RotCS *cs, int npairs, int K2, int ld, int nt,
                     bool priority = false) {
  switch (psd_blocks_per_lane(npairs, nt)) {
  case 1: update_all<1>(As, Ad, V, pq, cs, npairs, K2, ld, nt, priority); break;
  case 2: update_all<2>(As, Ad, V, pq, cs, npairs, K2, ld, nt, priority); break;
  case 3: update_all<3>(As, Ad, V, pq, cs, npairs, K2, ld, nt, priority); break;
  default: update_all<4>(As, Ad, V, pq, cs, npairs, K2, ld, nt, priority); break;
  }
}
// the entries the rotation wave of the signal form reads for pair (p, q) of step s + 1 must lie in PRIORITY blocks of step s (items
// 0 .. npri - 1 of the enumeration): returns the number of entries that do not
int priority_misses(const PsdPair *pairs_of_step, int npairs, int p, int q) {
  auto pair_of = [&](int player) {
    for (int i = 0; i < npairs; ++i)
      if (pairs_of_step[i].x == player || pairs_of_step[i].y == player) return i;
    return -1;
  };
  const int P = pair_of(p), Q = pair_of(q);
  const int npri = psd_priority_count(npairs);
  auto is_priority = [&](int a, int b) {
    if (a > b) std::swap(a, b);
    for (int e = 0; e < npri; ++e) {
      int bp, bq;
      psd_block_of_item(e, npairs, bp, bq);
      if (bp == a && bq == b) return true;
    }
    return false;
  };
  return (is_priority(P, Q) ? 0 : 1) + (is_priority(P, P) ? 0 : 1) + (is_priority(Q, Q) ? 0 : 1);
}
} // namespace

// Jacobi eigen-decomposition of the symmetric k x k matrix `a` (row major) with the kernel's schedule.
//   pipelined = 0: the two-phase step of rounds 2-4 (parameters, barrier, in-place update, barrier)
//   pipelined = 1: look-ahead + ping-pong (round 5)
//   pipelined = 2: signal form (round 6): priority blocks first, the rotation wave reads the updated entries
// out: evals[k] (diagonal of the rotated matrix), vecs[k*k] (row major, columns = eigenvectors), counters[0] = sweeps,
// [1] = rotating steps, [2] = steps, [3] = look-ahead source mismatches (must be 0: the players of pair i of step s+1 were found in
// the pairs the circle-method rule names).  Returns 0, or -1 if the sweep cap was hit.
extern "C" int psd_check_eig(const double *a, int k, int pipelined, double *evals, double *vecs, long *counters) {
  const int K2 = (k + 1) & ~1, npairs = K2 / 2, ld = K2 | 1;
  std::vector<real> A0((size_t)K2 * ld, 0.0), A1((size_t)K2 * ld, 0.0), V((size_t)K2 * ld, 0.0);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) A0[i * ld + j] = a[i * k + j];
  for (int i = 0; i < K2; ++i) V[i * ld + i] = 1;
  real fro = 0;
  for (int i = 0; i < K2; ++i)
    for (int j = 0; j < K2; ++j) fro += A0[i * ld + j] * A0[i * ld + j];
  fro = sqrt(fro);
  const real thr = 1e-15 * fro / (real)k;
  std::vector<PsdPair> tq(2 * TBL);
  std::vector<RotCS> tc(2 * TBL);
  int rot_any[2] = {0, 0};
  real *Acur = A0.data(), *Aoth = A1.data();
  long sweeps = 0, rsteps = 0, steps = 0, mism = 0;
  int sweep = 0;
  const int MAXS = 30;
  for (; sweep < MAXS && fro > 0; ++sweep) {
    real m = 0;
    for (int i = 0; i < K2; ++i)
      for (int j = i + 1; j < k; ++j) m = std::fmax(m, std::fabs(Acur[i * ld + j]));
    if (m <= thr) break;
    ++sweeps;
    real offmax = 0;
    std::vector<int> pa(npairs), pb(npairs);
    for (int i = 0; i < npairs; ++i) {
      pa[i] = i;
      pb[i] = K2 - 1 - i;
    }
    if (!pipelined) {
      for (int step = 0; step < K2 - 1; ++step, ++steps) {
        bool any = false;
        for (int i = 0; i < npairs; ++i) { // phase 1
          PsdRot r;
          any |= psd_first_rotation(Acur, pa[i], pb[i], ld, k, thr, offmax, r);
          psd_pair_advance(i, K2, pa[i], pb[i]);
          tq[i] = PsdPair{r.x, r.y};
          tc[i] = RotCS{r.c, r.s};
        }
        if (!any) continue; // barrier; uniform skip
        ++rsteps;
        update_dispatch(Acur, Acur, V.data(), tq.data(), tc.data(), npairs, K2, ld, NT_ALL); // phase 2, in place
      }
    } else if (pipelined == 2) {
      // signal form (round 6): the update (priority-first enumeration) writes the other copy; the rotation wave reads its three entries
      // from the UPDATED copy once the priority blocks are stored -- sequentially: after the update
      bool any = false;
      for (int i = 0; i < npairs; ++i) {
        PsdRot r;
        any |= psd_rotation_now(Acur, pa[i], pb[i], ld, k, thr, offmax, r);
        psd_pair_advance(i, K2, pa[i], pb[i]);
        tq[i] = PsdPair{r.x, r.y};
        tc[i] = RotCS{r.c, r.s};
      }
      rot_any[0] = any;
      for (int step = 0; step < K2 - 1; ++step, ++steps) {
        const int par = step & 1;
        const PsdPair *q0 = tq.data() + par * TBL;
        const RotCS *c0 = tc.data() + par * TBL;
        const bool rotates = rot_any[par] != 0;
        if (rotates) {
          ++rsteps;
          update_dispatch(Acur, Aoth, V.data(), q0, c0, npairs, K2, ld, NT_PIPE, true);
        }
        if (step + 1 < K2 - 1) {
          const real *Aread = rotates ? Aoth : Acur;
          bool nany = false;
          for (int i = 0; i < npairs; ++i) {
            mism += priority_misses(q0, npairs, pa[i], pb[i]); // what it reads was stored before the signal
            PsdRot r;
            nany |= psd_rotation_now(Aread, pa[i], pb[i], ld, k, thr, offmax, r);
            psd_pair_advance(i, K2, pa[i], pb[i]);
            tq[(par ^ 1) * TBL + i] = PsdPair{r.x, r.y};
            tc[(par ^ 1) * TBL + i] = RotCS{r.c, r.s};
          }
          rot_any[par ^ 1] = nany;
        }
        if (rotates) std::swap(Acur, Aoth); // after the barrier
      }
    } else {
      bool any = false;
      std::vector<PsdRot> mine(npairs), prev(npairs); // the look-ahead lanes' own registers: pair i of the current step
      for (int i = 0; i < npairs; ++i) { // prologue: step 0 from the matrix as it stands
        any |= psd_first_rotation(Acur, pa[i], pb[i], ld, k, thr, offmax, mine[i]);
        psd_pair_advance(i, K2, pa[i], pb[i]);
        tq[i] = PsdPair{mine[i].x, mine[i].y};
        tc[i] = RotCS{mine[i].c, mine[i].s};
      }
      rot_any[0] = any;
      for (int step = 0; step < K2 - 1; ++step, ++steps) {
        const int par = step & 1;
        const PsdPair *q0 = tq.data() + par * TBL;
        const RotCS *c0 = tc.data() + par * TBL;
        const bool rotates = rot_any[par] != 0;
        // look-ahead wave (reads Acur and its own registers' records of step `step`; writes the other tables)
        prev = mine;
        if (step + 1 < K2 - 1) {
          bool nany = false;
          for (int i = 0; i < npairs; ++i) {
            // the rule under test: the players of pair i of the next step sit in the pairs psd_lookahead reads
            const int sp = i == 0 ? 0 : (i == npairs - 1 ? npairs - 1 : i + 1), sq = i == 0 ? 1 : i - 1;
            if (!((q0[sp].x == pa[i] || q0[sp].y == pa[i]) && (q0[sq].x == pb[i] || q0[sq].y == pb[i]))) ++mism;
            // the kernel: lane i + 1's record by a DPP shift up, lane i - 1's by a shift down (lane 0 and the last lane use their own)
            const PsdRot &up = prev[i + 1 < npairs ? i + 1 : i], &dn = prev[i > 0 ? i - 1 : i];
            // (lane 0 names its players the other way round, so that every lane's first record is `up` -- the last lane's its own -- and its
            // second `dn`, which for lane 0 is its own record)
            const PsdRot &rec_p = i == npairs - 1 ? prev[i] : up, &rec_q = dn;
            const int p = i == 0 ? pb[i] : pa[i], q = i == 0 ? pa[i] : pb[i];
            if (i == 0 ? (&rec_p != &prev[sq] || &rec_q != &prev[sp]) : (&rec_p != &prev[sp] || &rec_q != &prev[sq])) ++mism;
            nany |= psd_lookahead_rec(Acur, rec_p, rec_q, p, q, ld, k, thr, offmax, mine[i]);
            psd_pair_advance(i, K2, pa[i], pb[i]);
            tq[(par ^ 1) * TBL + i] = PsdPair{mine[i].x, mine[i].y};
            tc[(par ^ 1) * TBL + i] = RotCS{mine[i].c, mine[i].s};
          }
          rot_any[par ^ 1] = nany;
        }
        if (rotates) {
          ++rsteps;
          update_dispatch(Acur, Aoth, V.data(), q0, c0, npairs, K2, ld, NT_PIPE);
          std::swap(Acur, Aoth); // after the barrier
        }
      }
    }
    if (offmax <= thr) {
      ++sweep;
      break;
    }
  }
  for (int i = 0; i < k; ++i) {
    evals[i] = Acur[i * ld + i];
    for (int j = 0; j < k; ++j) vecs[i * k + j] = V[i * ld + j];
  }
  counters[0] = sweeps;
  counters[1] = rsteps;
  counters[2] = steps;
  counters[3] = mism;
  return sweep >= MAXS ? -1 : 0;
}
