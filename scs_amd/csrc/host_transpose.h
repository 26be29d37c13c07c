// host_transpose.h -- the counting-sort transpose CSC(A) -> CSR(A) of the B1 boundary (scs_init_lin_sys_work is handed HOST arrays), as the
// reference does it in linsys/cpu/indirect/private.c:7-46.  From a few million entries on it runs on four threads and still returns the
// serial loop's result byte for byte (tests/native/host_check_transpose.cpp pins that on the CPU).  Included by linsys.hip only.
#pragma once
#include <algorithm>
#include <exception>
#include <system_error>
#include <thread>
#include <vector>

namespace scsamd {

// counting-sort transpose CSC(A) -> CSR(A) on the host (what private.c:7-46
// does); columns within each output row come out sorted.
template <typename eoff_t, typename real_t>
inline void host_transpose_t(int rows_out, int cols_out, const eoff_t *Ap, const int *Ai, const real_t *Ax,
                           std::vector<eoff_t> &Cp, std::vector<int> &Ci, std::vector<real_t> &Cx, bool force_serial = false,
                           long long parallel_from = 4000000) {
  // input: CSC with cols_out columns, rows_out rows.  output: CSR with rows_out rows.
  const long long nnz = Ap[cols_out];
  Cp.assign((size_t)rows_out + 1, 0);
  Ci.resize((size_t)nnz);
  Cx.resize((size_t)nnz);
  // From a few million entries on (the B1 boundary hands over host arrays; round 6's nnz = 2.2e9 run spent most of its 132 s of
  // scs_init_lin_sys_work in this loop): four ranges of columns with their own counts -- an entry's place is its row's start plus the
  // entries of that row in EARLIER columns, so the result is the serial loop's, byte for byte.
  constexpr int TT = 4;
  if (!force_serial && nnz >= parallel_from && cols_out >= 4 * TT) {
    int c0[TT + 1];
    for (int t = 0; t <= TT; ++t) c0[t] = t == TT ? cols_out : (int)(std::lower_bound(Ap, Ap + cols_out, (eoff_t)((double)nnz * t / TT)) - Ap);
    std::vector<std::vector<unsigned>> cnt(TT);
    std::exception_ptr err[TT];
    auto run = [&](auto fn) { // a std::bad_alloc inside a worker must reach the caller's catch, not std::terminate
      std::vector<std::thread> th;
      for (int t = 0; t < TT; ++t) err[t] = nullptr;
      auto body = [&](int t) {
        try {
          fn(t);
        } catch (...) {
          err[t] = std::current_exception();
        }
      };
      try {
        for (int t = 1; t < TT; ++t) th.emplace_back(body, t);
      } catch (const std::system_error &) {
        for (int t = (int)th.size() + 1; t < TT; ++t) body(t); // serial fallback for the ranges without a thread
      }
      body(0);
      for (std::thread &x : th) x.join();
      for (int t = 0; t < TT; ++t)
        if (err[t]) std::rethrow_exception(err[t]);
    };
    run([&](int t) {
      cnt[t].assign((size_t)rows_out, 0u);
      for (eoff_t k = Ap[c0[t]]; k < Ap[c0[t + 1]]; ++k) cnt[t][(size_t)Ai[k]]++;
    });
    for (int i = 0; i < rows_out; ++i) {
      eoff_t run_ = Cp[i];
      for (int t = 0; t < TT; ++t) {
        const unsigned c = cnt[t][(size_t)i];
        cnt[t][(size_t)i] = (unsigned)(run_ - Cp[i]);
        run_ += c;
      }
      Cp[(size_t)i + 1] = run_;
    }
    run([&](int t) {
      std::vector<unsigned> &off = cnt[t];
      for (int j = c0[t]; j < c0[t + 1]; ++j)
        for (eoff_t k = Ap[j]; k < Ap[j + 1]; ++k) {
          const eoff_t q = Cp[Ai[k]] + off[(size_t)Ai[k]]++;
          Ci[(size_t)q] = j;
          Cx[(size_t)q] = Ax[k];
        }
    });
    return;
  }
  for (long long k = 0; k < nnz; ++k) Cp[(size_t)Ai[k] + 1]++;
  for (int i = 0; i < rows_out; ++i) Cp[i + 1] += Cp[i];
  std::vector<eoff_t> nxt(Cp.begin(), Cp.end() - 1);
  for (int j = 0; j < cols_out; ++j)
    for (eoff_t k = Ap[j]; k < Ap[j + 1]; ++k) {
      const eoff_t q = nxt[Ai[k]]++;
      Ci[q] = j;
      Cx[q] = Ax[k];
    }
}

} // namespace scsamd
