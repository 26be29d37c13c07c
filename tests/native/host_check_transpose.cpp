// CPU pin of scs_amd/csrc/host_transpose.h: the four-thread transpose returns the serial counting sort's arrays byte for byte (ragged
// columns, empty columns and rows, duplicate-free sorted rows), for 32- and 64-bit entry positions and both precisions.  Test infrastructure.
#include "../../scs_amd/csrc/host_transpose.h"
#include <cstdio>
#include <cstring>
#include <random>
template <typename E, typename R>
static int check(unsigned seed, int rows, int cols, int per_col) {
  std::mt19937 rng(seed);
  std::vector<E> Ap(cols + 1, 0);
  std::vector<int> Ai;
  std::vector<R> Ax;
  for (int j = 0; j < cols; ++j) {
    std::vector<int> r;
    const int cnt = (rng() % 7 == 0) ? 0 : 1 + rng() % (2 * per_col);
    for (int k = 0; k < cnt; ++k) r.push_back(rng() % rows);
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());
    for (int v : r) {
      Ai.push_back(v);
      Ax.push_back((R)((int)(rng() % 2001) - 1000) / (R)1000);
    }
    Ap[j + 1] = (E)Ai.size();
  }
  std::vector<E> p0, p1;
  std::vector<int> i0, i1;
  std::vector<R> x0, x1;
  scsamd::host_transpose_t<E, R>(rows, cols, Ap.data(), Ai.data(), Ax.data(), p0, i0, x0, true);
  scsamd::host_transpose_t<E, R>(rows, cols, Ap.data(), Ai.data(), Ax.data(), p1, i1, x1, false, 1);
  if (p0 != p1 || i0 != i1 || x0.size() != x1.size() || memcmp(x0.data(), x1.data(), x0.size() * sizeof(R)) != 0) return 1;
  for (int r = 0; r < rows; ++r) // columns inside a row ascend (the reference's order)
    for (E k = p0[r] + 1; k < p0[r + 1]; ++k)
      if (i0[k] <= i0[k - 1]) return 2;
  return 0;
}
extern "C" int transpose_check(void) {
  int bad = 0;
  for (unsigned s = 0; s < 12; ++s) {
    bad |= check<int, double>(s, 500 + 37 * s, 300 + 11 * s, 6);
    bad |= check<long long, double>(100 + s, 2000, 40 + s, 25);
    bad |= check<int, float>(200 + s, 64, 4000, 3);
  }
  bad |= check<long long, double>(999, 30000, 20000, 12);
  return bad;
}
