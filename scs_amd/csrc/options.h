// options.h -- the ONE place this library's behaviour is steered from (round 6).
//
// The reference has no environment variables at all: everything a caller can change is a field of ScsSettings
// (include/scs.h:61-101).  Rounds 1-5 grew ~40 getenv() sites for A/B measurements and for tests that force a path; a stray
// SCS_AMD_* variable in the environment of a process that merely links this library changed numerics silently.  Now:
//
//   * every switch has ONE row in SCSAMD_OPTION_TABLE below -- key, class, whether it can change results beyond rounding,
//     one-line meaning.  INTEGRATION.md section 5 prints the same table; tests/test_options.py fails on any opt_*("key") in this
//     directory that has no row, on any row missing from INTEGRATION.md, and on any getenv outside this file;
//   * the programmatic entry is scs_amd_set_option("key", "value") / scs_amd_get_option("key") (include/scs_amd.h): process-wide,
//     read when a workspace is CREATED (scs_init, scs_init_lin_sys_work, _scs_init_cone / the first projection), never
//     during a solve; NULL value = back to the default;
//   * the environment is consulted only as the fallback SCS_AMD_<KEY IN CAPITALS>, and only for rows of class SUPPORTED or
//     DIAG unless SCS_AMD_ALLOW_ENV_HOOKS=1 is set (the test suites and the A/B scripts set it): AB and TEST rows -- the ones
//     that pick measurement variants or bend data structures -- cannot be reached by an inherited environment any more.
//
// Header-only (function-local statics): libscsamd_linsys.so links linsys.o alone, so there is no common object to put it in;
// inside one shared object the inline functions merge to a single table.
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>

namespace scsamd {

enum OptClass {
  OPT_SUPPORTED = 0, // a caller may legitimately want it; part of the interface
  OPT_AB = 1,        // measurement variant kept for A/B runs (profiles/): same algorithm, other schedule / kernel
  OPT_TEST = 2,      // test hook: forces a path or bends a data structure so a test can reach it
  OPT_DIAG = 3,      // diagnostics only (prints, trace files)
};

struct OptRow {
  const char *key;
  int cls;
  // 0: results identical bit for bit; 1: same mathematics, different summation / rotation order (rounding-level differences in
  // every iterate, O(CG tolerance) under the default inexact schedule); 2: changes the ALGORITHM's trajectory (iteration counts)
  int numerics;
  const char *values;
  const char *doc;
};

// clang-format off
#define SCSAMD_OPTION_TABLE(X) \
  X("reorder",          OPT_SUPPORTED, 1, "0|1",              "scs_init's internal renumbering of variables, zero/nonnegative rows and second-order-cone tails (callers never see it): 0 off, 1 = attempt and keep one even where the size screen would not bother (default: from 1e6 nonzeros on, a candidate is kept only if the measured line sharing of the gathers improves by 15-20 %)") \
  X("aa",               OPT_SUPPORTED, 1, "host|dev",         "Anderson acceleration on the host or on the device (default: device when n+m+1 >= 32768)") \
  X("equil",            OPT_SUPPORTED, 0, "host|dev",         "equilibration of scs_init on the host or on the device (default: device when nnz(A) >= 1e5; bit-identical)") \
  X("graph",            OPT_SUPPORTED, 0, "0|1",              "HIP-graph replay of blocks of 8 CG iterations for small systems (default on up to 2e6 nonzeros; 0 is needed under rocprofv3)") \
  X("psd_pipe",         OPT_SUPPORTED, 1, "0|1|2",            "Jacobi step of the LDS PSD kernel for orders <= 72: 1 (default) = pipelined with a look-ahead wave, 0 = two-phase step, 2 = pipelined in the signal form (the two-phase step's own rotations, one barrier per step; measured slower than 1)") \
  X("waverows",         OPT_SUPPORTED, 1, "0|1",              "wave-owned-rows SpMV layout (default: from 1e6 nonzeros on; 0 = CSR-stream kernel everywhere)") \
  X("psd_cold",         OPT_SUPPORTED, 1, "set",              "no eigenbasis carried between PSD projections (every projection starts cold)") \
  X("wr_lockstep",      OPT_AB,        1, "0|1|2",            "lockstep instantiation of the wave SpMV (default: fp64 from 5e6 nonzeros on); 2 = its chunk order with the plain kernel") \
  X("wr_ls_wpb",        OPT_AB,        0, "8|16",             "waves per workgroup of the lockstep kernel") \
  X("wr_ls_barriers",   OPT_AB,        0, "1|4",              "barriers per chunk of the lockstep kernel (default: from the measured line sharing)") \
  X("wr_ls_order",      OPT_AB,        1, "0|1",              "quarter-window chunk order of the lockstep layout") \
  X("wr_wpc",           OPT_AB,        1, "1..16",            "resident waves per CU the wave layout is cut for") \
  X("wr_nnz",           OPT_AB,        1, "n",                "nonzero budget per unit of the wave layout") \
  X("wr_pipe",          OPT_AB,        0, "0|1|2",            "instantiation of the plain wave SpMV: 1 = next chunk's stream in flight ahead of the gathers, 2 = two chunks per round trip (default: from the size and the measured line sharing)") \
  X("reorder_home",     OPT_AB,        1, "0|1|2|3|4",        "chain + home numbering, how the movable rows are placed: 4 (default) = grouped by a hashed home column, line-sized groups dealt wide (blocked stride); 1 = plain home order; 0 = home is the first column; 2 = rows stay (chain only); 3 = homes only, columns as given") \
  X("reorder_block",    OPT_AB,        1, "rows",             "chain + home numbering, blocked stride: rows per block (default: one 128-byte line of the gathered vector)") \
  X("reorder_stride",   OPT_AB,        1, "blocks",           "chain + home numbering, blocked stride: blocks per stream (default 32; 0 = sqrt(blocks) streams)") \
  X("vec_nt",           OPT_AB,        0, "0|1|3",            "non-temporal policy of the CG vector kernels") \
  X("dir_mode",         OPT_AB,        0, "n",                "variant of k_cg_direction") \
  X("cg3",              OPT_AB,        1, "0|1",              "three launches per CG iteration (k_cg3_update; default off)") \
  X("psd_blocked",      OPT_AB,        1, "0|1",              "blocked tournament Jacobi for PSD orders > 92 (0 = round 2's single-column steps)") \
  X("psd_cross",        OPT_AB,        1, "0|1",              "cross sweeps of the blocked Jacobi subproblem (0 = full 63-step sweeps)") \
  X("psd_fused",        OPT_AB,        0, "0|1",              "one fused launch per outer step of the blocked Jacobi iteration (0 = two launches)") \
  X("psd_offscan",      OPT_AB,        0, "0|1|2",            "PSD orders > 92: whether another sweep would rotate anything is read off the matrix as the sweep leaves it: 1 (default) = by the sweep's last update itself, 2 = by a pass of its own (k_bp_offscan), 0 = found out by running the sweep (rounds 2-5)") \
  X("psd_grid",         OPT_AB,        0, "0|1",              "PSD orders > 92, fused step: 1 (default) = one-dimensional grid with the inner-sweep workgroups of all blocks first, 0 = (job, block) grid of rounds 4-5") \
  X("psd_prologue",     OPT_AB,        0, "0|1",              "PSD orders > 92, fused step: 1 (default) = the inner sweep's subproblem from ONE level of global loads (both candidates of every entry beside the flags that pick), 0 = rounds 4-5's three dependent levels") \
  X("psd_warm_kmax",    OPT_AB,        1, "k",                "largest LDS PSD order that is warm started (72 = round 3's gate)") \
  X("fused",            OPT_TEST,      1, "0|1",              "force / forbid the one-workgroup PCG of tiny systems") \
  X("cg2",              OPT_TEST,      1, "0|1",              "force / forbid the two-launch CG iteration (n <= 1024)") \
  X("box_multi",        OPT_TEST,      1, "0|1",              "force the chip-wide / one-workgroup box-cone Newton iteration") \
  X("vec_max_grid",     OPT_TEST,      1, "g",                "cap on the grid of the vector kernels (tests force grid-striding)") \
  X("spmv_max_grid",    OPT_TEST,      0, "g",                "cap on the grid of the CSR-stream kernel") \
  X("wr_wide",          OPT_TEST,      0, "0|1",              "the wide wave layout (column word + 16-bit local rows): forced at any size (tests), and the only way to get a wave layout beyond 2^26 rows or columns (default there: CSR-stream kernel, same speed, 61 GB less HBM at nnz = 2.2e9)") \
  X("wr_build",         OPT_TEST,      0, "dev|host|verify",  "who builds the wave layout; verify builds both and fails scs_init on any differing byte") \
  X("transpose",        OPT_TEST,      0, "dev|host|verify",  "who builds the pattern transpose; verify as above") \
  X("test_offset_bias", OPT_TEST,      0, "b",                "DLONG build: every stored entry position + b, arrays shifted back (64-bit positions without a 26 GB matrix)") \
  X("debug",            OPT_DIAG,      0, "set",              "setup phase timings and per-solve PCG / PSD sweep counts on stderr") \
  X("trace_file",       OPT_DIAG,      0, "path",             "append every scs_solve_lin_sys call's inputs / outputs (oracle/trace_linsys.c's record)")
// clang-format on

inline const OptRow *opt_rows(int *count) {
#define X(k, c, n, v, d) {k, c, n, v, d},
  static const OptRow rows[] = {SCSAMD_OPTION_TABLE(X)};
#undef X
  if (count) *count = (int)(sizeof(rows) / sizeof(rows[0]));
  return rows;
}
inline const OptRow *opt_find(const char *key) {
  int n;
  const OptRow *r = opt_rows(&n);
  for (int i = 0; i < n; ++i)
    if (!strcmp(r[i].key, key)) return &r[i];
  return nullptr;
}

// Values are INTERNED: every distinct string ever set (or read from the environment) is kept for the life of the process in a
// node-based container, and the tables map a key to a pointer into it.  A pointer returned by opt_get therefore never dangles, whatever
// another host thread sets meanwhile (batch runs create workspaces from several threads); the pool grows by one short string per
// distinct value, i.e. by nothing in practice.
struct OptState {
  std::mutex mu;
  std::set<std::string> pool;                // interned values (std::set never moves its elements)
  std::map<std::string, const char *> set;   // programmatic values
};
inline OptState &opt_state() {
  static OptState s;
  return s;
}

// 0 on success, -1 on an unknown key.  value == NULL removes the programmatic value (back to environment / default).
inline int opt_set(const char *key, const char *value) {
  if (!key || !opt_find(key)) return -1;
  OptState &s = opt_state();
  std::lock_guard<std::mutex> g(s.mu);
  if (value) s.set[key] = s.pool.insert(value).first->c_str();
  else s.set.erase(key);
  return 0;
}

// the value in force, or NULL when the option is at its default.  The returned pointer stays valid for the life of the process.
inline const char *opt_get(const char *key) {
  const OptRow *row = opt_find(key);
  if (!row) return nullptr; // unknown keys never reach the environment (tests/test_options.py keeps the table complete)
  OptState &s = opt_state();
  std::lock_guard<std::mutex> g(s.mu);
  auto it = s.set.find(key);
  if (it != s.set.end()) return it->second;
  if (row->cls == OPT_AB || row->cls == OPT_TEST) {
    const char *allow = getenv("SCS_AMD_ALLOW_ENV_HOOKS");
    if (!allow || !atoi(allow)) return nullptr;
  }
  std::string name = "SCS_AMD_";
  for (const char *c = key; *c; ++c) name += (char)((*c >= 'a' && *c <= 'z') ? *c - 'a' + 'A' : *c);
  const char *e = getenv(name.c_str());
  if (!e) return nullptr;
  return s.pool.insert(e).first->c_str(); // (the environment's own storage may be rewritten by setenv / putenv: a copy is returned)
}
inline bool opt_is_set(const char *key) { return opt_get(key) != nullptr; }

} // namespace scsamd
