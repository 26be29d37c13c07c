"""Small random inputs for kernel-level parity tests (numpy only)."""
import numpy as np
import scipy.sparse as sp


def random_csc(m, n, col_nnz, seed, dtype=np.float64):
    """m x n CSC with exactly col_nnz distinct sorted rows per column, values U[-1,1]
    (same structure law as reference test/problem_utils.h:64-79)."""
    rng = np.random.default_rng(seed)
    rows = np.empty((n, col_nnz), dtype=np.int64)
    for j in range(n):
        rows[j] = np.sort(rng.choice(m, size=col_nnz, replace=False))
    vals = rng.uniform(-1, 1, size=(n, col_nnz)).astype(dtype)
    indptr = np.arange(0, (n + 1) * col_nnz, col_nnz, dtype=np.int32)
    return sp.csc_matrix((vals.ravel(), rows.ravel().astype(np.int32), indptr), shape=(m, n))


def diag_r(n, m, z, rho_x=1e-6, scale=0.1, dtype=np.float64):
    """[R_x; R_y] as built by reference src/scs.c:971-980 + src/cones.c:349-363."""
    d = np.empty(n + m, dtype=dtype)
    d[:n] = rho_x
    d[n:n + z] = 1.0 / (1000.0 * scale)
    d[n + z:] = 1.0 / scale
    return d
