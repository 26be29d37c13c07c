// reorder.cpp -- see reorder.h.  Host code, setup time only (scs_init).
#include "reorder.h"
#include <thread>
#include <functional>
#include <exception>
#include <system_error>
#include <atomic>
#include <algorithm>
#include <chrono>
#include <numeric>

namespace scsamd {

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A piece of work on a side thread that cannot take the process down: an exception inside the work (std::bad_alloc of a
// measurement's scratch at n = 1e6) is caught there and rethrown by join() on the joining thread -- where scs_init's catch turns it
// into a NULL workspace, as before the threads existed; a thread that cannot be created (pid / thread limits) runs the work inline;
// the destructor joins without throwing, so unwinding past a task in flight never reaches std::terminate through ~thread.
class SideTask {
  std::thread th;
  std::function<void()> fn;
  std::exception_ptr err;
  void run() noexcept {
    try {
      fn();
    } catch (...) {
      err = std::current_exception();
    }
  }

public:
  SideTask() = default;
  SideTask(const SideTask &) = delete;
  SideTask &operator=(const SideTask &) = delete;
  void start(std::function<void()> f) {
    fn = std::move(f);
    try {
      th = std::thread([this] { run(); });
    } catch (const std::system_error &) {
      run(); // serial fallback
    }
  }
  void join() {
    if (th.joinable()) th.join();
    if (err) {
      std::exception_ptr e = err;
      err = nullptr;
      std::rethrow_exception(e);
    }
  }
  ~SideTask() {
    if (th.joinable()) th.join();
  }
};

constexpr int TR_THREADS = 4; // fixed (results do not depend on it, only the time does)

// transpose of a pattern: (ptr, idx) with `rows` rows over `cols` columns -> (tptr, tidx) with `cols` rows (counting sort,
// like linsys/cpu/indirect/private.c:7-46 without the values).  From a million entries on the two passes over the entries run on
// TR_THREADS ranges of source rows with per-range counts; an entry's place is its row's start + the entries of the same output row in
// EARLIER source rows, so the result is the serial counting sort's, byte for byte.
void transpose_pattern(const eoff *ptr, const int *idx, int rows, int cols, std::vector<eoff> &tptr, std::vector<int> &tidx) {
  const size_t nnz = (size_t)ptr[rows];
  if (nnz >= 1000000 && rows >= 4 * TR_THREADS) {
    int r0[TR_THREADS + 1];
    for (int t = 0; t <= TR_THREADS; ++t) { // ranges of source rows with about equal numbers of entries
      const eoff want = (eoff)((double)nnz * t / TR_THREADS);
      r0[t] = t == TR_THREADS ? rows : (int)(std::lower_bound(ptr, ptr + rows, want) - ptr);
    }
    std::vector<std::vector<unsigned>> cnt(TR_THREADS);
    {
      SideTask th[TR_THREADS - 1];
      auto count = [&](int t) {
        cnt[t].assign((size_t)cols, 0u);
        for (eoff k = ptr[r0[t]]; k < ptr[r0[t + 1]]; ++k) cnt[t][(size_t)idx[k]]++;
      };
      for (int t = 1; t < TR_THREADS; ++t) th[t - 1].start([&count, t] { count(t); });
      count(0);
      for (SideTask &x : th) x.join();
    }
    tptr.assign((size_t)cols + 1, 0);
    for (int c = 0; c < cols; ++c) { // per-range counts -> per-range first positions (in place), row starts
      eoff run = tptr[c];
      for (int t = 0; t < TR_THREADS; ++t) {
        const unsigned k = cnt[t][(size_t)c];
        cnt[t][(size_t)c] = (unsigned)(run - tptr[c]); // offset inside the output row: < 2^32 (a row of A or A' is far shorter)
        run += k;
      }
      tptr[(size_t)c + 1] = run;
    }
    tidx.resize(nnz);
    {
      SideTask th[TR_THREADS - 1];
      auto fill = [&](int t) {
        std::vector<unsigned> &off = cnt[t];
        for (int r = r0[t]; r < r0[t + 1]; ++r)
          for (eoff k = ptr[r]; k < ptr[r + 1]; ++k) tidx[(size_t)(tptr[idx[k]] + off[(size_t)idx[k]]++)] = r;
      };
      for (int t = 1; t < TR_THREADS; ++t) th[t - 1].start([&fill, t] { fill(t); });
      fill(0);
      for (SideTask &x : th) x.join();
    }
    return;
  }
  tptr.assign((size_t)cols + 1, 0);
  for (size_t k = 0; k < nnz; ++k) tptr[(size_t)idx[k] + 1]++;
  for (int c = 0; c < cols; ++c) tptr[c + 1] += tptr[c];
  tidx.resize(nnz);
  std::vector<eoff> fill(tptr.begin(), tptr.end() - 1);
  for (int r = 0; r < rows; ++r)
    for (eoff k = ptr[r]; k < ptr[r + 1]; ++k) tidx[(size_t)fill[idx[k]]++] = r;
}

// stable argsort of keys; entries with key < 0 ("no key") keep their relative order behind the keyed ones
std::vector<int> order_by_key(const std::vector<double> &key) {
  std::vector<int> ord(key.size());
  std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
    const bool ka = key[a] >= 0, kb = key[b] >= 0;
    if (ka != kb) return ka;
    return ka && key[a] < key[b];
  });
  return ord;
}

} // namespace

static void build_renumbered(const HostCsc &A, const std::vector<int> &col_new2old, const std::vector<int> &row_new2old, HostCsc &B);

double lines_per_entry(const eoff *ptr, const int *idx, int rows, int cols, size_t elem_bytes) {
  const long long nnz = ptr[rows];
  if (nnz <= 0) return 1.0;
  const int lshift = elem_bytes == 8 ? 4 : 5; // 128-byte line = 16 fp64 / 32 fp32 entries
  const long long budget = std::max<long long>(1024, nnz / 2048); // ~ the unit spmv_wave.h gives a wave on the whole chip
  std::vector<int> stamp(((size_t)cols >> lshift) + 2, -1);
  long long distinct = 0, acc = 0;
  int unit = 0, unit_rows = 0;
  for (int r = 0; r < rows; ++r) {
    const long long rn = ptr[r + 1] - ptr[r];
    if ((acc + rn > budget && unit_rows > 0) || unit_rows >= 1024) {
      ++unit;
      acc = 0;
      unit_rows = 0;
    }
    for (eoff k = ptr[r]; k < ptr[r + 1]; ++k) {
      int &st = stamp[(size_t)idx[k] >> lshift];
      if (st != unit) {
        st = unit;
        ++distinct;
      }
    }
    acc += rn;
    ++unit_rows;
  }
  return (double)distinct / (double)nnz;
}

namespace {

struct Candidate {
  std::vector<int> col_new2old, row_new2old;
  double after[2] = {1, 1};
  const char *method = "";
};

// free rows ([0, z) and [z, z + l)) sorted inside their range by `rkey` (< 0: no key, kept behind); all other rows stay
void place_free_rows(const std::vector<double> &rkey, int z, int lp, int m, std::vector<int> &row_new2old) {
  row_new2old.resize((size_t)m);
  std::iota(row_new2old.begin(), row_new2old.end(), 0);
  for (int range = 0; range < 2; ++range) {
    const int a = range == 0 ? 0 : z, b = range == 0 ? z : z + lp;
    if (b - a < 2) continue;
    const std::vector<double> key(rkey.begin() + a, rkey.begin() + b);
    const std::vector<int> ord = order_by_key(key);
    for (int t = 0; t < b - a; ++t) row_new2old[a + t] = a + ord[t];
  }
}

void measure(const HostCsc &A, Candidate &c) {
  const int m = A.m, n = A.n;
  const eoff *cp = A.p.data();
  const int *ci = A.i.data();
  std::vector<int> row_old2new((size_t)m);
  for (int i = 0; i < m; ++i) row_old2new[c.row_new2old[i]] = i;
  std::vector<eoff> np((size_t)n + 1, 0);
  std::vector<int> ni((size_t)cp[n]);
  for (int j = 0; j < n; ++j) np[j + 1] = np[j] + (cp[c.col_new2old[j] + 1] - cp[c.col_new2old[j]]);
  auto fill = [&](int j0, int j1) {
    for (int j = j0; j < j1; ++j) {
      eoff o = np[j];
      const int jo = c.col_new2old[j];
      for (eoff q = cp[jo]; q < cp[jo + 1]; ++q) ni[(size_t)o++] = row_old2new[ci[q]];
    }
  };
  {
    SideTask th[TR_THREADS - 1];
    for (int t = 1; t < TR_THREADS; ++t) {
      const int j0 = (int)((long long)n * t / TR_THREADS), j1 = (int)((long long)n * (t + 1) / TR_THREADS);
      th[t - 1].start([&fill, j0, j1] { fill(j0, j1); });
    }
    fill(0, (int)((long long)n / TR_THREADS));
    for (SideTask &x : th) x.join();
  }
  std::vector<eoff> tp;
  std::vector<int> ti;
  // (the two measurements are independent passes over read-only patterns: one of them on a second host thread)
  SideTask side; // (declared after everything its work reads: joined before those go away, also on unwinding)
  side.start([&] { c.after[1] = lines_per_entry(np.data(), ni.data(), n, m, sizeof(real)); });
  transpose_pattern(np.data(), ni.data(), n, m, tp, ti);
  c.after[0] = lines_per_entry(tp.data(), ti.data(), m, n, sizeof(real));
  side.join();
}

// ---- candidate 3 (round 6): "chain + home" numbering for patterns WITHOUT hidden locality (the benchmark family: every column's rows
// uniformly random).  Nothing can make most gathers of such a matrix share lines -- its row / column graph is an expander -- but a
// fixed fraction can be had by construction:
//   * chain: the columns are numbered along greedy walks in which consecutive columns share a row (every step uses a row not used
//     as a link before).  The two entries of that row then sit in neighbouring columns: ONE line of x serves both gathers of the A
//     product, and in the A' product the neighbouring columns -- rows of A', handled by the same wave back to back -- ask for the
//     same y entry.  One entry per column turns local.
//   * home: every row that may move (zero cone, nonnegative cone, the tail of a second-order cone: |x|_2 does not depend on the order
//     of x -- src/cones.c:1247-1279 -- and the equilibration keeps D constant inside a cone, linsys/scs_matrix.c:257,329) is keyed by
//     ONE of its columns (picked by a hash of the row); inside its cone's range the rows are ordered by that key, cut into blocks of one
//     line's worth of rows, and the blocks are dealt wide (blocked stride, below).  Rows with neighbouring homes share a line of x in
//     the A product, and a column finds the rows that call it home in one line of y in the A' product.  One entry per row turns local.
// On n = 1e6, m = 2e6, 10 per column that is 3e6 of 1e7 entries: measured 0.96 / 0.98 -> 0.77 / 0.72 distinct lines per entry.
// Deterministic: the walks run on HOME_THREADS fixed column ranges (a walk never leaves its range), whatever the machine has.
constexpr int HOME_THREADS = 4;

// the ranges of rows that may be renumbered among themselves: [0, z), [z, z + l), and behind the box cone the tail (all but the first
// row) of every second-order cone (row order of the cones: include/scs.h:121-172)
void movable_ranges(const ScsCone *k, int m, std::vector<std::pair<int, int>> &rg) {
  rg.clear();
  long long o = 0;
  if (k->z > 0) rg.emplace_back(0, (int)k->z);
  o += k->z;
  if (k->l > 0) rg.emplace_back((int)o, (int)(o + k->l));
  o += k->l;
  o += k->bsize;
  for (scs_int i = 0; i < k->qsize && k->q; ++i) {
    if (k->q[i] > 2 && o + k->q[i] <= m) rg.emplace_back((int)o + 1, (int)(o + k->q[i]));
    o += k->q[i];
  }
}

void chain_home_candidate(const HostCsc &A, const ScsCone *k, const std::vector<eoff> &rptr, const std::vector<int> &rcol, Candidate &c) {
  const int m = A.m, n = A.n;
  const eoff *cp = A.p.data();
  const int *ci = A.i.data();
  c.method = "chain + home (columns along walks that share a row; movable rows grouped by a home column, line-sized groups dealt wide)";
  // ---- chain: greedy walks, one set per fixed column range
  std::vector<unsigned char> visited((size_t)n, 0);
  std::vector<std::vector<int>> part(HOME_THREADS);
  auto walk = [&](int t) {
    const int c0 = (int)((long long)n * t / HOME_THREADS), c1 = (int)((long long)n * (t + 1) / HOME_THREADS);
    std::vector<unsigned char> used((size_t)m, 0); // per walk set: a row that linked two columns, or has no unvisited column left in the range
    std::vector<int> &ord = part[t];
    ord.reserve((size_t)(c1 - c0));
    int nxt = c0;
    for (;;) {
      while (nxt < c1 && visited[nxt]) ++nxt;
      if (nxt >= c1) break;
      int cur = nxt;
      for (;;) {
        visited[cur] = 1; // (only this thread touches [c0, c1))
        ord.push_back(cur);
        int found = -1;
        for (eoff q = cp[cur]; q < cp[cur + 1] && found < 0; ++q) {
          const int r = ci[q];
          if (used[r]) continue;
          for (eoff e = rptr[r]; e < rptr[r + 1]; ++e) {
            const int c2 = rcol[e];
            if (c2 >= c0 && c2 < c1 && !visited[c2]) {
              found = c2;
              break;
            }
          }
          used[r] = 1; // links cur -> found, or is exhausted for this range (visited only grows)
        }
        if (found < 0) break;
        cur = found;
      }
    }
  };
  {
    SideTask th[HOME_THREADS - 1];
    for (int t = 1; t < HOME_THREADS; ++t) th[t - 1].start([&walk, t] { walk(t); });
    walk(0);
    for (SideTask &x : th) x.join();
  }
  const bool dbg = opt_get("debug") != nullptr;
  const double tw = now_s();
  c.col_new2old.clear();
  c.col_new2old.reserve((size_t)n);
  for (int t = 0; t < HOME_THREADS; ++t) c.col_new2old.insert(c.col_new2old.end(), part[t].begin(), part[t].end());
  if (const char *e = opt_get("reorder_home"))
    if (atoi(e) == 3) std::iota(c.col_new2old.begin(), c.col_new2old.end(), 0); // measurements: homes only, columns as given
  std::vector<int> colpos((size_t)n);
  for (int j = 0; j < n; ++j) colpos[c.col_new2old[j]] = j;
  // ---- home: one column of every row
  // Which of its columns a row calls home: ONE PICKED BY A HASH OF THE ROW, not the first in the new order.  With the first column
  // (measured, profiles/r6_chain_home.md) a row's other entries all lie to the right of its home, so a unit of rows late in the
  // order gathers from a compressed range of x and falls out of step with the window of x the rest of the chip is gathering from
  // (spmv_wave.h keeps that window L2-resident): the A product went from 64.8 to 72.1 us although its L1 -> L2 requests fell by
  // 19 %.  A hashed choice keeps a unit's entries uniform over the columns and the homes uniform over the rows.
  int home_mode = 4; // 4 (default): hashed homes in blocked stride; measurements: 0 first column, 1 hashed column in plain home order, 2 rows stay (chain only), 3 hashed homes with the columns as given
  if (const char *e = opt_get("reorder_home")) home_mode = atoi(e);
  std::vector<int> home((size_t)m, n); // n = a row without entries: behind the others
  auto homes = [&](int r0, int r1) {
    for (int r = r0; r < r1; ++r) {
      const eoff a = rptr[r], b = rptr[r + 1];
      int h = n;
      if (home_mode == 0) {
        for (eoff e = a; e < b; ++e) h = std::min(h, colpos[rcol[e]]);
      } else if (b > a) {
        const unsigned hsh = ((unsigned)r * 2654435761u) >> 8;
        h = colpos[rcol[a + (eoff)(hsh % (unsigned)(b - a))]];
      }
      home[r] = h;
    }
  };
  {
    SideTask th[HOME_THREADS - 1];
    for (int t = 1; t < HOME_THREADS; ++t) {
      const int r0 = (int)((long long)m * t / HOME_THREADS), r1 = (int)((long long)m * (t + 1) / HOME_THREADS);
      th[t - 1].start([&homes, r0, r1] { homes(r0, r1); });
    }
    homes(0, (int)((long long)m / HOME_THREADS));
    for (SideTask &x : th) x.join();
  }
  const double th_ = now_s();
  // ---- movable rows in home order inside their range: one counting sort by home over all rows (stable in the row index), then
  // every row is dealt to the next free place of its range
  std::vector<std::pair<int, int>> rg;
  movable_ranges(k, m, rg);
  std::vector<int> range_of((size_t)m, -1);
  for (size_t g = 0; g < rg.size(); ++g)
    for (int r = rg[g].first; r < rg[g].second; ++r) range_of[r] = (int)g;
  std::vector<int> cnt((size_t)n + 2, 0);
  for (int r = 0; r < m; ++r)
    if (range_of[r] >= 0) cnt[(size_t)home[r] + 1]++;
  for (int h = 0; h <= n; ++h) cnt[(size_t)h + 1] += cnt[h];
  std::vector<int> sorted((size_t)cnt[(size_t)n + 1]);
  for (int r = 0; r < m; ++r)
    if (range_of[r] >= 0) sorted[(size_t)cnt[home[r]]++] = r;
  c.row_new2old.resize((size_t)m);
  std::iota(c.row_new2old.begin(), c.row_new2old.end(), 0);
  std::vector<int> cursor(rg.size());
  for (size_t g = 0; g < rg.size(); ++g) cursor[g] = rg[g].first;
  if (home_mode != 2)
    for (int r : sorted) c.row_new2old[(size_t)cursor[range_of[r]]++] = r;
  if (home_mode == 4) {
    // Blocked stride (round 6, measured; profiles/r6_chain_home.md): rows in plain home order make a unit of ~500 consecutive rows put
    // a fifth of its entries on ONE spot of x -- the A product then pays more L2 misses than the shared lines save (65.5 -> 67.5 us
    // while the A' product gains 12; homes alone, without the chain: 69.7).  What both products need is only that rows with
    // neighbouring homes share a LINE: blocks of one line's worth of rows stay together, and the blocks of a range are dealt to
    // streams of `per_stream` blocks, so that consecutive blocks are far apart in home order and a unit's homes spread over all of
    // x.  A product 67.5 -> 57.1 us, A' 55.5 -> 56.1 us, same request counts.
    int B = (int)(128 / sizeof(real)), per_stream = 32; // (measured: 8 ... 64 blocks per stream within 1 %, sqrt(blocks) streams 1.5 % behind)
    if (const char *e = opt_get("reorder_block")) B = std::max(1, atoi(e));
    if (const char *e = opt_get("reorder_stride")) per_stream = std::max(0, atoi(e)); // blocks per stream (0: sqrt(blocks) streams)
    std::vector<int> tmp;
    for (size_t g = 0; g < rg.size(); ++g) {
      const int a = rg[g].first, len = rg[g].second - a;
      const int nb = (len + B - 1) / B;
      if (nb < 4) continue;
      int S = 1;
      if (per_stream > 0) S = std::max(1, nb / per_stream);
      else
        while ((long long)S * S < nb) ++S;
      tmp.assign(c.row_new2old.begin() + a, c.row_new2old.begin() + a + len);
      int o = a;
      for (int st = 0; st < S; ++st)
        for (int b = st; b < nb; b += S)
          for (int t = b * B; t < std::min(len, (b + 1) * B); ++t) c.row_new2old[(size_t)o++] = tmp[(size_t)t];
    }
  }
  if (dbg) fprintf(stderr, "[scs_amd reorder] chain + home: walks done at +0, homes %.0f ms, row placement %.0f ms\n", 1e3 * (th_ - tw), 1e3 * (now_s() - th_));
}

} // namespace

void plan_reorder(const HostCsc &A, const ScsCone *k, bool has_P, Reorder &R) {
  const double t0 = now_s();
  R = Reorder();
  const int m = A.m, n = A.n;
  const long long nnz = n > 0 ? A.p[n] : 0;
  int force = -1;
  if (const char *e = opt_get("reorder")) force = atoi(e);
  if (force == 0) {
    R.why = "SCS_AMD_REORDER=0";
    return;
  }
  if (has_P) {
    R.why = "P present";
    return;
  }
  if (nnz < 1000000 && force != 1) { // below the wave-owned-rows layout's threshold the gathered vectors sit in L2 anyway
    R.why = "fewer than 1e6 nonzeros";
    return;
  }
  if (m < 2 || n < 2) return;
  // beyond 2^26 rows or columns the packed word of the wave-owned-rows layout has no room for the column (spmv_wave.h, col_bits): the
  // products run through the CSR-stream kernel, and a decision that costs a minute of host time at nnz = 2e9 would buy nothing
  if ((m > (1 << 26) || n > (1 << 26)) && force != 1) {
    R.why = "more than 2^26 rows or columns (no wave-owned-rows layout to help)";
    return;
  }
  const int z = (int)k->z, lp = (int)k->l, fixed0 = z + lp; // rows [0, z) and [z, z + l) may move inside their range; the rest are anchors
  const eoff *cp = A.p.data();
  const int *ci = A.i.data();
  // ---- one pass: anchors per column and how tightly they sit
  std::vector<double> ckey((size_t)n, -1.0);
  long long anchored = 0, spread_cols = 0, unkeyed = 0;
  double spread_sum = 0;
  const double fixed_len = std::max(1, m - fixed0);
  for (int j = 0; j < n; ++j) {
    double s = 0;
    int cnt = 0, lo = m, hi = -1;
    for (eoff q = cp[j]; q < cp[j + 1]; ++q) {
      const int i = ci[q];
      if (i >= fixed0) {
        s += i;
        ++cnt;
        lo = std::min(lo, i);
        hi = std::max(hi, i);
      }
    }
    if (cnt) {
      ckey[j] = s / cnt;
      anchored += cnt;
      if (cnt >= 2) {
        spread_sum += (hi - lo) / fixed_len;
        ++spread_cols;
      }
    } else {
      ++unkeyed;
    }
  }
  const bool dbg_early = opt_get("debug") != nullptr;
  const bool many_anchors = anchored * 5 >= nnz; // a fifth of the entries sit in rows that cannot move
  if (many_anchors) { // (also when the attempt is forced: a pattern without hidden locality goes to candidate 3 either way)
    // k anchors drawn uniformly from the fixed rows span (k - 1) / (k + 1) of them on average (0.6 - 0.7 on the benchmark family);
    // a hidden band spans a sliver.  Nothing to recover from a uniformly random pattern: say so after this one pass.
    const double mean_spread = spread_cols ? spread_sum / (double)spread_cols : 1.0;
    if (mean_spread > 0.25 && sizeof(real) == 4 && force != 1) {
      // fp32 (configs[4], n = 4e6), measured: in blocked stride the product gains (275 -> 256 us) but scs_init grows by 2.7 s and the
      // one to-eps run under the numbering needed 525 iterations instead of 350 (fp32 sits close to its rounding floor at eps = 1e-3 and
      // the other summation order moved it): 22.4 + 3.1 s against 20.7 + 0.4 s.  Not attempted in the fp32 build.
      R.why = "no hidden locality (fp32 build: the chain + home numbering is not attempted, see reorder.cpp)";
      R.seconds = now_s() - t0;
      return;
    }
    if (mean_spread > 0.25) {
      // no hidden locality to recover (rounds 4-5 stopped here): what CAN be had by construction is candidate 3 (round 6)
      std::vector<eoff> rptr0;
      std::vector<int> rcol0;
      SideTask tb1;
      tb1.start([&] { R.before[1] = lines_per_entry(cp, ci, n, m, sizeof(real)); });
      transpose_pattern(cp, ci, n, m, rptr0, rcol0);
      SideTask tb2;
      tb2.start([&] { R.before[0] = lines_per_entry(rptr0.data(), rcol0.data(), m, n, sizeof(real)); });
      Candidate c3;
      if (dbg_early) fprintf(stderr, "[scs_amd reorder] first pass + transpose: %.0f ms\n", 1e3 * (now_s() - t0));
      chain_home_candidate(A, k, rptr0, rcol0, c3);
      if (dbg_early) fprintf(stderr, "[scs_amd reorder] chain + home numbering built at %.0f ms\n", 1e3 * (now_s() - t0));
      // the renumbered matrix is built BESIDE the measurement (the candidate is kept on this family; a rejected one costs nothing but
      // the side threads' time): 0.43 -> 0.33 s of scs_init at n = 1e6
      HostCsc built;
      {
        SideTask tbuild; // (joined before `built` is used or goes away)
        tbuild.start([&] { build_renumbered(A, c3.col_new2old, c3.row_new2old, built); });
        measure(A, c3);
        tbuild.join();
      }
      tb2.join();
      tb1.join();
      R.after[0] = c3.after[0];
      R.after[1] = c3.after[1];
      R.method = c3.method;
      const double before = 0.5 * (R.before[0] + R.before[1]), after = 0.5 * (c3.after[0] + c3.after[1]);
      if (after <= 0.85 * before || force == 1) { // (forced attempts keep the candidate whatever it measures: A/B runs of its parts)
        R.active = true;
        R.col_new2old = std::move(c3.col_new2old);
        R.row_new2old = std::move(c3.row_new2old);
        R.ready = std::move(built);
        R.have_ready = true;
        R.why = "no hidden locality (anchored entries spread over the whole cone range); chain + home numbering shares 15 % or more of the gathers' lines";
      } else {
        R.why = "no hidden locality, and the chain + home numbering does not share 15 % of the gathers' lines";
      }
      R.seconds = now_s() - t0;
      if (dbg_early)
        fprintf(stderr, "[scs_amd reorder] %s: lines/entry A %.3f -> %.3f, A' %.3f -> %.3f, %s (%.0f ms)\n", R.method, R.before[0], R.after[0], R.before[1],
                R.after[1], R.active ? "kept" : "dropped", 1e3 * R.seconds);
      return;
    }
  }
  std::vector<eoff> rptr; // CSR pattern of A (rows -> columns)
  std::vector<int> rcol;
  std::atomic<int> verdict{0}; // 0: not known yet, 1: go on, 2: the given numbering is already local
  SideTask tb0, tb, t1; // destroyed (joined) before everything above; what t1 writes (c1) is declared below but outlives the joins: see c1
  tb0.start([&] { R.before[1] = lines_per_entry(cp, ci, n, m, sizeof(real)); }); // A' product: rows = columns of A, gathers y
  transpose_pattern(cp, ci, n, m, rptr, rcol);
  // The line sharing of the GIVEN numbering is measured on a side thread while this one already starts on the candidates; if it turns
  // out local enough (the common case for problems that come in a sensible order) the searches are told to stop and their work --
  // at most the ~0.15 s the measurement takes -- is dropped.
  // (0.25, not the 0.8 at which spmv_wave.h switches kernels: rows that keep their place re-use a few columns many times and
  // so share lines in ANY numbering of the variables -- a scrambled band measures 0.49 / 0.94 -- while the rest gains 10x)
  tb.start([&] {
    struct Always { // whatever happens in here, the searches must not wait for a verdict that never comes
      std::atomic<int> &v;
      ~Always() {
        int zero = 0;
        v.compare_exchange_strong(zero, 1, std::memory_order_release);
      }
    } always{verdict};
    R.before[0] = lines_per_entry(rptr.data(), rcol.data(), m, n, sizeof(real)); // A product: rows of A, gathers x
    tb0.join();
    verdict.store(0.5 * (R.before[0] + R.before[1]) <= 0.25 && force != 1 ? 2 : 1, std::memory_order_release);
  });
  // The candidates are independent of each other (read-only A, rptr / rcol, ckey): candidate 1 is built and measured on a second
  // host thread while this one runs the graph search of candidate 2 (round 5: 1.17 -> ~0.6 s of scs_init at n = 1e6, same decisions).
  const bool dbg_t = opt_get("debug") != nullptr;
  if (dbg_t) fprintf(stderr, "[scs_amd reorder] first pass + transpose: %.0f ms\n", 1e3 * (now_s() - t0));
  std::vector<Candidate> cands;
  Candidate c1; // written by t1: every path below joins t1 before c1 is read or goes out of scope (JoinFirst for the unwinding path)
  struct JoinFirst {
    SideTask &t;
    ~JoinFirst() {
      try {
        t.join();
      } catch (...) {
      }
    }
  } c1_guard{t1};
  bool have_c1 = false;
  // ---- candidate 1: anchors.  A column is keyed by the mean position of its entries in rows that cannot move (only when every
  // column has such entries: the rest would need the graph search anyway), a free row by the mean NEW position of its columns.
  if (many_anchors && unkeyed == 0) {
    have_c1 = true;
    t1.start([&] {
    Candidate &c = c1;
    c.method = "anchors (mean position of a column's entries in the rows that cannot move)";
    c.col_new2old = order_by_key(ckey);
    std::vector<int> col_old2new((size_t)n);
    for (int j = 0; j < n; ++j) col_old2new[c.col_new2old[j]] = j;
    std::vector<double> rkey((size_t)m, -1.0);
    for (int i = 0; i < fixed0; ++i) {
      double s = 0;
      int cnt = 0;
      for (eoff q = rptr[i]; q < rptr[i + 1]; ++q) {
        s += col_old2new[rcol[q]];
        ++cnt;
      }
      if (cnt) rkey[i] = s / cnt;
    }
    place_free_rows(rkey, z, lp, m, c.row_new2old);
    if (verdict.load(std::memory_order_acquire) != 2) measure(A, c);
    });
  }
  // ---- candidate 2: breadth-first (Cuthill-McKee) numbering of the bipartite graph (vertices 0..n-1 = columns, n..n+m-1 = rows),
  // every connected component from a pseudo-peripheral start; columns take the visiting order, free rows follow it inside their
  // ranges, anchored rows stay (a band is walked end to end, so the anchored rows see their columns in a moving window too)
  {
    Candidate c;
    c.method = "Cuthill-McKee (breadth-first numbering of the row / column graph)";
    const size_t nv = (size_t)n + m;
    std::vector<int> mark(nv, -1), order, scratch;
    order.reserve(nv);
    scratch.reserve(nv);
    const int *rcol_p = rcol.data();
    auto bfs = [&](int start, int tag, std::vector<int> &q) -> int { // appends the component of `start` to q; returns the last vertex reached
      size_t head = q.size();
      q.push_back(start);
      mark[start] = tag;
      int last = start;
      while (head < q.size()) {
        if ((head & 0xFFFF) == 0 && verdict.load(std::memory_order_relaxed) == 2) return -1; // told to stop (see above)
        // the search is bound by the latency of its scattered reads (one of `mark` per edge): the marks of the neighbours of the vertex
        // two places down the queue, and the adjacency of the one six places down, are asked for now
        if (head + 6 < q.size()) {
          const int w = q[head + 6];
          __builtin_prefetch(w < n ? (const void *)(ci + cp[w]) : (const void *)(rcol_p + rptr[w - n])); // raw pointers: an empty row at the end points one past the array
        }
        if (head + 2 < q.size()) {
          const int w = q[head + 2];
          if (w < n)
            for (eoff e = cp[w]; e < cp[w + 1]; ++e) __builtin_prefetch(&mark[(size_t)n + ci[e]]);
          else
            for (eoff e = rptr[w - n]; e < rptr[w - n + 1]; ++e) __builtin_prefetch(&mark[rcol[e]]);
        }
        const int v = q[head++];
        last = v;
        if (v < n) {
          for (eoff e = cp[v]; e < cp[v + 1]; ++e) {
            const int u = n + ci[e];
            if (mark[u] != tag) {
              mark[u] = tag;
              q.push_back(u);
            }
          }
        } else {
          const int i = v - n;
          for (eoff e = rptr[i]; e < rptr[i + 1]; ++e) {
            const int u = rcol[e];
            if (mark[u] != tag) {
              mark[u] = tag;
              q.push_back(u);
            }
          }
        }
      }
      return last;
    };
    std::vector<char> done(nv, 0);
    int tag = 0;
    for (size_t s = 0; s < nv; ++s) {
      if (done[s]) continue;
      scratch.clear();
      if (verdict.load(std::memory_order_relaxed) == 2) break;
      const int far = bfs((int)s, tag++, scratch); // pseudo-peripheral start: the last vertex a search from s reaches (a band is walked to its
                                                   // farther end; a second search from there measured the same line sharing and a quarter more time)
      if (far < 0) break;
      const size_t first = order.size();
      if (bfs(far, tag++, order) < 0) break;
      for (size_t q = first; q < order.size(); ++q) done[order[q]] = 1;
    }
    tb.join();
    if (verdict.load(std::memory_order_acquire) == 2) {
      if (have_c1) t1.join();
      R.why = "the given numbering is already local";
      R.seconds = now_s() - t0;
      return;
    }
    // every vertex is visited (all components are walked): the free rows take the visiting order inside their ranges directly -- what
    // place_free_rows does with the visiting rank as the key, without the two sorts (180 ms of 1.17 s at n = 1e6)
    c.col_new2old.reserve(n);
    c.row_new2old.resize((size_t)m);
    std::iota(c.row_new2old.begin(), c.row_new2old.end(), 0);
    size_t o0 = 0, o1 = (size_t)z;
    for (size_t q = 0; q < order.size(); ++q) {
      if (order[q] < n) {
        c.col_new2old.push_back(order[q]);
      } else {
        const int i = order[q] - n;
        if (i < z) c.row_new2old[o0++] = i;
        else if (i < z + lp) c.row_new2old[o1++] = i;
      }
    }
    if (dbg_t) fprintf(stderr, "[scs_amd reorder] graph search + row placement done at %.0f ms\n", 1e3 * (now_s() - t0));
    measure(A, c);
    if (dbg_t) fprintf(stderr, "[scs_amd reorder] candidate 2 measured at %.0f ms\n", 1e3 * (now_s() - t0));
    if (have_c1) { // (candidate 1 first, as before: ties go to it)
      t1.join();
      cands.push_back(std::move(c1));
    }
    cands.push_back(std::move(c));
  }
  size_t best = 0;
  for (size_t q = 1; q < cands.size(); ++q)
    if (cands[q].after[0] + cands[q].after[1] < cands[best].after[0] + cands[best].after[1]) best = q;
  Candidate &c = cands[best];
  R.after[0] = c.after[0];
  R.after[1] = c.after[1];
  R.method = c.method;
  R.seconds = now_s() - t0;
  const double before = 0.5 * (R.before[0] + R.before[1]); // (tb was joined after the graph search)
  if (0.5 * (c.after[0] + c.after[1]) <= 0.8 * before) {
    R.active = true;
    R.col_new2old = std::move(c.col_new2old);
    R.row_new2old = std::move(c.row_new2old);
    R.why = "line sharing of the gathers improved by 20 % or more";
  } else if (sizeof(real) == 8 && before > 0.8) {
    // nothing to recover by the graph search either (e.g. a uniformly random LP: no anchors, so the one-pass screen above cannot tell):
    // the construction of candidate 3, as for the patterns the screen does recognise
    Candidate c3;
    chain_home_candidate(A, k, rptr, rcol, c3);
    measure(A, c3);
    if (0.5 * (c3.after[0] + c3.after[1]) <= 0.85 * before) {
      R.active = true;
      R.after[0] = c3.after[0];
      R.after[1] = c3.after[1];
      R.method = c3.method;
      R.col_new2old = std::move(c3.col_new2old);
      R.row_new2old = std::move(c3.row_new2old);
      R.why = "no hidden locality found by the graph search; chain + home numbering shares 15 % or more of the gathers' lines";
    } else {
      R.why = "no candidate numbering improved the measured line sharing";
    }
    R.seconds = now_s() - t0;
  } else {
    R.why = "no candidate numbering improved the measured line sharing by 20 %";
  }
  if (opt_get("debug"))
    fprintf(stderr, "[scs_amd reorder] %s: lines/entry A %.3f -> %.3f, A' %.3f -> %.3f, %s (%.0f ms)\n", R.method, R.before[0], R.after[0],
            R.before[1], R.after[1], R.active ? "kept" : "dropped", 1e3 * R.seconds);
}

// B <- A[row_new2old, col_new2old], row indices sorted inside every column
static void build_renumbered(const HostCsc &A, const std::vector<int> &col_new2old, const std::vector<int> &row_new2old, HostCsc &B) {
  const int m = A.m, n = A.n;
  std::vector<int> row_old2new((size_t)m);
  for (int i = 0; i < m; ++i) row_old2new[row_new2old[i]] = i;
  B.m = m;
  B.n = n;
  B.p.assign((size_t)n + 1, 0);
  B.i.resize(A.i.size());
  B.x.resize(A.x.size());
  for (int j = 0; j < n; ++j) B.p[j + 1] = B.p[j] + (A.p[col_new2old[j] + 1] - A.p[col_new2old[j]]);
  // columns are independent once B.p is known: ranges of them on a few host threads
  auto do_range = [&](int j0, int j1) {
    std::vector<std::pair<int, real>> col;
    for (int j = j0; j < j1; ++j) {
      const int jo = col_new2old[j];
      col.clear();
      for (eoff q = A.p[jo]; q < A.p[jo + 1]; ++q) col.emplace_back(row_old2new[A.i[q]], A.x[q]);
      std::stable_sort(col.begin(), col.end(), [](const std::pair<int, real> &a, const std::pair<int, real> &b) { return a.first < b.first; });
      eoff o = B.p[j];
      for (const auto &e : col) {
        B.i[(size_t)o] = e.first;
        B.x[(size_t)o] = e.second;
        ++o;
      }
    }
  };
  const int nthr = n >= 100000 ? 4 : 1;
  SideTask pool[3];
  for (int t = 1; t < nthr; ++t) {
    const int j0 = (int)((long long)n * t / nthr), j1 = (int)((long long)n * (t + 1) / nthr);
    pool[t - 1].start([&do_range, j0, j1] { do_range(j0, j1); });
  }
  do_range(0, (int)((long long)n / nthr));
  for (SideTask &th : pool) th.join();
}

void apply_reorder(HostCsc &A, const Reorder &R) {
  if (!R.active) return;
  if (R.have_ready) { // built by plan_reorder beside its measurement
    A = std::move(R.ready);
    R.have_ready = false;
    return;
  }
  HostCsc B;
  build_renumbered(A, R.col_new2old, R.row_new2old, B);
  A = std::move(B);
}

} // namespace scsamd
