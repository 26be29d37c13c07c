// Host-only sanitizer driver (AddressSanitizer + UndefinedBehaviorSanitizer, and ThreadSanitizer in a second build) for the numbering decision of
// scs_amd/csrc/reorder.cpp: plan_reorder + apply_reorder on random patterns with mixed cones (one of them large enough for the threaded
// transposes and walks), permutations checked.  Test infrastructure; built by tests/test_reorder.py with hipcc --cuda-host-only (GPU sanitizers are
// not available on the pool).
#include "../../scs_amd/csrc/reorder.cpp"
#include <random>
using namespace scsamd;
int main() {
  std::mt19937 rng(7);
  for (int trial = 0; trial < 40; ++trial) {
    const int n = trial == 0 ? 150000 : 50 + rng() % 4000, cn = trial == 0 ? 10 : 2 + rng() % 9;
    std::vector<long long> q;
    int z = trial == 0 ? 30000 : rng() % 200, l = trial == 0 ? 90000 : rng() % 400, nb = (rng() % 3) ? 0 : 1 + rng() % 30;
    int nq = rng() % 12;
    long long m = z + l + (nb ? nb + 1 : 0);
    for (int i = 0; i < nq; ++i) { q.push_back(trial == 0 ? 15000 + rng() % 100 : 1 + rng() % 300); m += q.back(); }
    int s3 = rng() % 3; m += 6 * s3; // PSD blocks of order 3
    if (m < 4) continue;
    HostCsc A; A.m = (int)m; A.n = n; A.p.assign(n + 1, 0);
    for (int j = 0; j < n; ++j) {
      std::vector<int> r;
      for (int k = 0; k < cn; ++k) r.push_back(rng() % m);
      std::sort(r.begin(), r.end()); r.erase(std::unique(r.begin(), r.end()), r.end());
      if (rng() % 50 == 0) r.clear(); // an empty column now and then
      for (int v : r) { A.i.push_back(v); A.x.push_back((real)1); }
      A.p[j + 1] = (eoff)A.i.size();
    }
    ScsCone k{}; std::vector<scs_int> qq(q.begin(), q.end()), ss(s3, 3);
    std::vector<scs_float> bu(nb, 1), bl(nb, -1);
    k.z = z; k.l = l; k.bsize = nb ? nb + 1 : 0; k.bu = nb ? bu.data() : nullptr; k.bl = nb ? bl.data() : nullptr;
    k.q = qq.empty() ? nullptr : qq.data(); k.qsize = (scs_int)qq.size(); k.s = ss.empty() ? nullptr : ss.data(); k.ssize = s3;
    Reorder R;
    plan_reorder(A, &k, false, R);
    HostCsc B = A;
    apply_reorder(B, R);
    if (B.i.size() != A.i.size() || B.p[n] != A.p[n]) { printf("size mismatch\n"); return 1; }
    if (R.active) {
      std::vector<char> seen(n, 0);
      for (int v : R.col_new2old) { if (v < 0 || v >= n || seen[v]) { printf("bad col perm\n"); return 1; } seen[v] = 1; }
      std::vector<char> sr(m, 0);
      for (int v : R.row_new2old) { if (v < 0 || v >= m || sr[v]) { printf("bad row perm\n"); return 1; } sr[v] = 1; }
    }
  }
  printf("sanitizer driver ok\n");
  return 0;
}
