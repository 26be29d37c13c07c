"""Kernel-level measurement of the B1 path on one GPU: PCG on a random
n x (2n) system at the headline size; prints CG iterations/s and the sampled
HIP-event SpMV time -> algorithmic GB/s (SURVEY.md section 8d byte count)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scs_amd import capi, problems  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--m", type=int, default=0)
    ap.add_argument("--col-nnz", type=int, default=10)
    ap.add_argument("--tol", type=float, default=1e-9)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lib", default="libscsamd_linsys.so")
    a = ap.parse_args()
    n, m = a.n, a.m or 2 * a.n
    lib = capi.load(a.lib)
    T = lib._scs_types
    rng = np.random.default_rng(0)
    t0 = time.time()
    rows = problems.random_rows(m, n, a.col_nnz, rng)
    vals = rng.uniform(-1, 1, size=(n, a.col_nnz)).astype(T.np_float)
    import scipy.sparse as sp
    A = sp.csc_matrix((vals.ravel(), rows.ravel().astype(np.int32),
                       np.arange(0, (n + 1) * a.col_nnz, a.col_nnz, dtype=np.int32)), shape=(m, n))
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m), T=T)
    dr = np.empty(n + m, dtype=T.np_float)
    dr[:n] = 1e-6
    dr[n:n + m // 10] = 1.0 / 100.0
    dr[n + m // 10:] = 10.0
    print(f"gen {time.time()-t0:.1f}s n={n} m={m} nnz={A.nnz}", flush=True)
    t0 = time.time()
    w = lib.scs_init_lin_sys_work(C.byref(prob.matA), None, dr.ctypes.data_as(T.fp))
    assert w
    print(f"init {time.time()-t0:.2f}s", flush=True)
    lib.scs_amd_linsys_set_profiling(w, 1)
    b0 = rng.uniform(-1, 1, n + m).astype(T.np_float)
    st = T.ScsAmdStats()
    prev_its = 0
    for r in range(a.reps):
        b = b0.copy()
        t0 = time.time()
        rc = lib.scs_solve_lin_sys(w, b.ctypes.data_as(T.fp), None, a.tol)
        dt = time.time() - t0
        lib.scs_amd_linsys_get_stats(w, C.byref(st))
        its = st.cg_iters - prev_its
        prev_its = st.cg_iters
        print(json.dumps(dict(rep=r, rc=rc, cg_its=its, wall_s=round(dt, 4), cg_its_per_s=round(its / dt, 1),
                              us_per_cg_it=round(1e6 * dt / max(its, 1), 2))), flush=True)
    spmv_avg_ms = st.spmv_ms / max(st.spmv_launches, 1)
    bytes_per_spmv = st.spmv_bytes / 2
    print(json.dumps(dict(spmv_samples=st.spmv_launches, spmv_avg_us=round(1e3 * spmv_avg_ms, 2),
                          spmv_alg_MB=round(bytes_per_spmv / 1e6, 1),
                          spmv_GBps=round(bytes_per_spmv / (spmv_avg_ms * 1e-3) / 1e9, 1),
                          cg_ms_total=round(st.cg_ms, 2))), flush=True)
    lib.scs_free_lin_sys_work(w)


if __name__ == "__main__":
    main()
