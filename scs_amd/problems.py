"""Synthetic problem generators for the benchmark configurations.

`random_cone_prob` restates the law of reference test/problem_utils.h:22-81
(`gen_random_prob_data`) and the cone recipe of test/random_socp_prob.c:79-107:
draw z ~ U[-1,1]^m, y = Proj_{K*}(z), s = y - z (so s in K, y in K*, s'y = 0),
x ~ U[-1,1]^n, A with exactly `col_nnz` distinct rows per column and U[-1,1]
values, b = A x + s, c = -A' y.  The reference's loop is O(n*m) RNG draws
(problem_utils.h:64-79) and unusable at n = 1e6; here rows are sampled with a
vectorised draw-sort-redraw, so the DATA differ from the reference generator's
while the distribution is the same.  (Bit-identical reference data for the small
plumbing config is produced by oracle/ for the tests.)

Pure numpy/scipy, host side; the cone projection used to place y in K* is a
numpy restatement for zero / nonnegative / box(t only) / SOC / PSD cones.
"""
import math

import numpy as np
import scipy.sparse as sp


def socp_cone_sizes(m, p_f=0.1, p_l=0.3, rand=None, q_fixed=None):
    """Cone split of test/random_socp_prob.c:79-107: z = floor(m p_f), l = floor(m p_l),
    SOC sizes (rand() % max_q) + 1 with max_q = ceil(m / log m) until the rows are
    used up.  `rand` is a callable returning non-negative ints (libc rand() in the
    reference); defaults to a fixed-seed numpy stream.  `q_fixed` gives the
    "many small cones" variant (all SOCs of that size)."""
    z = int(math.floor(m * p_f))
    l = int(math.floor(m * p_l))
    rows = m - z - l
    q = []
    if q_fixed:
        while rows > 0:
            sz = min(q_fixed, rows)
            q.append(sz)
            rows -= sz
    else:
        if rand is None:
            rs = np.random.RandomState(12345)
            rand = lambda: int(rs.randint(0, 2 ** 31 - 1))
        max_q = int(math.ceil(m / math.log(m)))
        while rows > max_q:
            sz = rand() % max_q + 1
            q.append(sz)
            rows -= sz
        if rows > 0:
            q.append(rows)
    return dict(z=z, l=l, q=q)


def proj_dual_cone_np(v, cone):
    """Euclidean projection onto the dual cone K* (numpy; generator use only).
    Moreau: Proj_{K*}(v) = v + Proj_K(-v)   (reference src/cones.c:1552-1596)."""
    v = np.asarray(v, dtype=np.float64)
    w = -v.copy()
    i = 0
    z = cone.get("z", 0)
    w[i:i + z] = 0.0
    i += z
    l = cone.get("l", 0)
    np.maximum(w[i:i + l], 0.0, out=w[i:i + l])
    i += l
    bu = np.asarray(cone.get("bu", []), dtype=np.float64)
    if len(bu):
        bl = np.asarray(cone["bl"], dtype=np.float64)
        w[i:i + len(bu) + 1] = _proj_box_np(w[i:i + len(bu) + 1], bl, bu)
        i += len(bu) + 1
    for q in cone.get("q", []):
        w[i:i + q] = _proj_soc_np(w[i:i + q])
        i += q
    for k in cone.get("s", []):
        sz = k * (k + 1) // 2
        w[i:i + sz] = _proj_psd_np(w[i:i + sz], k)
        i += sz
    assert i == len(v), (i, len(v))
    return v + w


def _proj_soc_np(x):
    if len(x) == 1:
        return np.maximum(x, 0.0)
    t, nrm = x[0], np.linalg.norm(x[1:])
    if nrm <= t:
        return x
    if nrm <= -t:
        return np.zeros_like(x)
    a = 0.5 * (nrm + t)
    out = x * (a / nrm)
    out[0] = a
    return out


def _proj_box_np(tx, bl, bu):
    t, x = tx[0], tx[1:]
    tt = max(t, 0.0)
    for _ in range(100):  # Newton on t (src/cones.c:1208-1234), Euclidean metric
        hi, lo = x > tt * bu, x < tt * bl
        g = (tt - t) + np.sum((tt * bu[hi] - x[hi]) * bu[hi]) + np.sum((tt * bl[lo] - x[lo]) * bl[lo])
        h = 1.0 + np.sum(bu[hi] ** 2) + np.sum(bl[lo] ** 2)
        tn = max(tt - g / max(h, 1e-8), 0.0)
        if abs(tn - tt) < 1e-13 * max(tn, 1.0):
            tt = tn
            break
        tt = tn
    return np.concatenate([[tt], np.clip(x, tt * bl, tt * bu)])


def _proj_psd_np(v, k):
    X = np.zeros((k, k))
    idx = np.tril_indices(k)
    # packed lower triangle, column major (src/cones.c:1020): iterate columns
    cols, rows = np.triu_indices(k)  # (col <= row) pairs in column-major order of the lower tri
    X[rows, cols] = v
    X = X + X.T - np.diag(np.diag(X))
    off = ~np.eye(k, dtype=bool)
    X[off] /= math.sqrt(2.0)
    w, V = np.linalg.eigh(X)
    Xp = (V * np.maximum(w, 0.0)) @ V.T
    Xp[off] *= math.sqrt(2.0)
    return Xp[rows, cols]


def random_rows(m, n, col_nnz, rng):
    """n x col_nnz int array: distinct sorted rows per column, uniform over [0, m)."""
    r = rng.integers(0, m, size=(n, col_nnz), dtype=np.int64)
    r.sort(axis=1)
    while True:
        dup = np.any(r[:, 1:] == r[:, :-1], axis=1)
        nd = int(dup.sum())
        if nd == 0:
            break
        r[dup] = np.sort(rng.integers(0, m, size=(nd, col_nnz), dtype=np.int64), axis=1)
    return r


def banded_rows(m, n, col_nnz, band, rng):
    """n x col_nnz int array: distinct sorted rows per column, uniform over a window of `band` rows centred on the
    column's own position j m / n (clipped to [0, m)) -- a matrix with the column locality of staged / time-indexed
    models (MPC, trajectory, multi-period portfolio), where a variable only meets constraints of nearby stages."""
    band = int(min(max(band, 2 * col_nnz), m))
    lo = (np.arange(n, dtype=np.int64) * m) // n - band // 2
    lo = np.clip(lo, 0, m - band)[:, None]
    r = lo + rng.integers(0, band, size=(n, col_nnz), dtype=np.int64)
    r.sort(axis=1)
    while True:
        dup = np.any(r[:, 1:] == r[:, :-1], axis=1)
        nd = int(dup.sum())
        if nd == 0:
            break
        r[dup] = np.sort(lo[dup] + rng.integers(0, band, size=(nd, col_nnz), dtype=np.int64), axis=1)
    return r


def scramble_prob(pr, seed):
    """The same problem in an arbitrary numbering: variables (columns of A, c) randomly permuted, rows randomly permuted INSIDE
    the zero and the nonnegative cone (rows of A, b); every other cone keeps its rows (their order is part of the cone,
    include/scs.h:121-172).  Returns the scrambled problem plus `col_perm` / `row_perm` (new index -> original index)."""
    rng = np.random.default_rng(seed)
    A = pr["A"]
    m, n = A.shape
    cone = pr["cone"]
    z, l = int(cone.get("z", 0)), int(cone.get("l", 0))
    colp = rng.permutation(n)
    rowp = np.arange(m)
    rowp[:z] = rng.permutation(z)
    rowp[z:z + l] = z + rng.permutation(l)
    As = sp.csc_matrix(A[rowp][:, colp])
    As.sort_indices()
    out = dict(pr)
    out.update(A=sp.csc_matrix((As.data.astype(A.dtype), As.indices.astype(np.int32), As.indptr.astype(np.int32)), shape=(m, n)),
               b=pr["b"][rowp], c=pr["c"][colp], col_perm=colp, row_perm=rowp)
    for key, perm in (("x_opt", colp), ("y_opt", rowp), ("s_opt", rowp)):
        if key in pr:
            out[key] = pr[key][perm]
    return out


def random_cone_prob(n, m, col_nnz, cone, seed=1234, dtype=np.float64, band=None):
    """Feasible & bounded random cone program.  Returns dict(A (csc), b, c, cone,
    x_opt, y_opt, s_opt).  `band` (rows): column-local sparsity pattern (banded_rows) instead of the
    reference generator's uniform one."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(-1, 1, m)
    y = proj_dual_cone_np(z, cone)
    s = y - z
    x = rng.uniform(-1, 1, n)
    rows = banded_rows(m, n, col_nnz, band, rng) if band else random_rows(m, n, col_nnz, rng)
    vals = rng.uniform(-1, 1, size=(n, col_nnz))
    indptr = np.arange(0, (n + 1) * col_nnz, col_nnz, dtype=np.int64)
    A = sp.csc_matrix((vals.ravel(), rows.ravel(), indptr), shape=(m, n))
    b = A @ x + s
    c = -(A.T @ y)
    A = sp.csc_matrix((A.data.astype(dtype), A.indices.astype(np.int32), A.indptr.astype(np.int32)),
                      shape=(m, n))
    return dict(A=A, b=b.astype(dtype), c=c.astype(dtype), cone=cone, x_opt=x, y_opt=y, s_opt=s)


def random_socp(n, m=None, col_nnz=None, seed=1234, q_fixed=None, dtype=np.float64, band=None, scramble=None):
    """The headline family: LP + SOC cones only (BASELINE.json configs 1, 2, 4, 5).  `band`: the same cones and
    data law on a column-local (banded) pattern -- the locality variant of bench.py's `secondary`.  `scramble` (a seed):
    that problem handed over in an arbitrary numbering of its variables and zero / nonnegative rows (scramble_prob)."""
    m = 2 * n if m is None else m
    col_nnz = 10 if col_nnz is None else col_nnz
    cone = socp_cone_sizes(m, q_fixed=q_fixed)
    pr = random_cone_prob(n, m, col_nnz, cone, seed=seed, dtype=dtype, band=band)
    return scramble_prob(pr, scramble) if scramble is not None else pr


def random_sdp(n, n_blocks=200, block=50, bsize=1001, col_nnz=10, seed=1234):
    """BASELINE.json config 3: `n_blocks` PSD cones of size block x block + a box cone."""
    cone = dict(z=0, l=0, bu=np.ones(bsize - 1), bl=-np.ones(bsize - 1), bsize=bsize, q=[],
                s=[block] * n_blocks)
    m = bsize + n_blocks * (block * (block + 1) // 2)
    return random_cone_prob(n, m, col_nnz, cone, seed=seed)
