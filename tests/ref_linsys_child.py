"""Child process of tests/test_fullsize_gpu.py: ONE scs_solve_lin_sys of the reference's OpenMP flavour (oracle/_ref,
linsys/cpu/indirect/private.c:284-324) on a generated problem, in a process of its own so that OMP_NUM_THREADS /
OMP_WAIT_POLICY are read by a fresh libgomp.  Test infrastructure only (CPU).

    python tests/ref_linsys_child.py n m col_nnz seed z tol out.npy
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def inputs(n, m, col_nnz, seed, z):
    from scs_amd import capi, problems
    from tests import probgen
    pr = problems.random_socp(n, m, col_nnz, seed=seed)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    dr = probgen.diag_r(n, m, z=z)
    rng = np.random.default_rng(5)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n) * 0.1
    return prob, dr, b, s


if __name__ == "__main__":
    n, m, col_nnz, seed, z = (int(v) for v in sys.argv[1:6])
    tol, out = float(sys.argv[6]), sys.argv[7]
    from oracle import pyoracle
    ref = pyoracle.load_ref("libscsindir_ref_omp.so")
    T = ref._scs_types
    prob, dr, b, s = inputs(n, m, col_nnz, seed, z)
    w = ref.scs_init_lin_sys_work(C.byref(prob.matA), None, dr.ctypes.data_as(T.fp))
    assert w
    o = b.copy()
    t0 = time.time()
    assert ref.scs_solve_lin_sys(w, o.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp), tol) == 0
    print("reference solve %.1f s" % (time.time() - t0), flush=True)
    ref.scs_free_lin_sys_work(w)
    np.save(out, o)
