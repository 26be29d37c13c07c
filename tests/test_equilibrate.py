"""Data equilibration (reference linsys/scs_matrix.c:433-496) through the
scs_amd_equilibrate hook.  CPU: the host code against the reference's own normalised
matrix (golden fixture dumped from the reference build).  GPU: the device kernels are
bit-identical to the host code, with and without P, over every cone shape that changes
the per-cone aggregation."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from scs_amd import capi, problems

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _equilibrate(lib, A, P, cone, where):
    T = capi.T64
    A = sp.csc_matrix(A)
    A.sort_indices()
    m, n = A.shape
    Ax = np.ascontiguousarray(A.data, dtype=np.float64).copy()
    Ai = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    Am = T.ScsMatrix(Ax.ctypes.data_as(dp), Ai.ctypes.data_as(ip), Ap.ctypes.data_as(ip), m, n)
    Pm, Px, keep = None, None, []
    if P is not None:
        P = sp.csc_matrix(sp.triu(P))
        P.sort_indices()
        Px = np.ascontiguousarray(P.data, dtype=np.float64).copy()
        Pi = np.ascontiguousarray(P.indices, dtype=np.int32)
        Pp = np.ascontiguousarray(P.indptr, dtype=np.int32)
        keep = [Pi, Pp]
        Pm = T.ScsMatrix(Px.ctypes.data_as(dp), Pi.ctypes.data_as(ip), Pp.ctypes.data_as(ip), n, n)
    k = capi.make_cone(cone, T)
    D, E = np.zeros(m), np.zeros(n)
    lib.scs_amd_equilibrate.restype = C.c_int
    lib.scs_amd_equilibrate.argtypes = [C.POINTER(T.ScsMatrix), C.POINTER(T.ScsMatrix), C.POINTER(T.ScsCone), dp, dp,
                                        C.c_int]
    rc = lib.scs_amd_equilibrate(C.byref(Am), C.byref(Pm) if Pm is not None else None, C.byref(k),
                                 D.ctypes.data_as(dp), E.ctypes.data_as(dp), where)
    assert rc == 0
    del keep
    return Ax, Px, D, E


def test_host_equilibration_matches_reference_normalised_matrix():
    g = np.load(os.path.join(G, "linsys_cfg1.npz"))
    pr = problems.random_socp(1000, 3000, 32, seed=1234)
    lib = capi.load("libscsamd.so")
    Ax, _, D, E = _equilibrate(lib, pr["A"], None, pr["cone"], 0)
    want = g["Ax_normalized"]
    assert Ax.shape == want.shape
    # same passes in the same order; the reference build may contract a*b+c, ours does not
    np.testing.assert_allclose(Ax, want, rtol=1e-13, atol=0)
    assert D.min() > 0 and E.min() > 0
    # the equilibrated matrix is diag(D) A diag(E)
    A = sp.csc_matrix(pr["A"])
    A.sort_indices()
    ref = (sp.diags(D) @ A @ sp.diags(E)).tocsc()
    ref.sort_indices()
    np.testing.assert_allclose(Ax, ref.data, rtol=1e-12)


def _mixed_cone_matrix(seed):
    """Every cone type that has its own equilibration segment; A is just random sparse."""
    cone = dict(z=5, l=7, bu=[1.0, 2.0, 3.0], bl=[-1.0, -2.0, 0.0], q=[3, 70, 1, 200], s=[4, 1, 13], cs=[3],
                ep=2, ed=1, p=[0.3, -0.6])
    m, n = capi.cone_rows(cone), 120
    rng = np.random.default_rng(seed)
    rows = problems.random_rows(m, n, 9, rng)
    vals = rng.uniform(-1, 1, size=(n, 9)) * rng.choice([1e-3, 1.0, 50.0], size=(n, 1))
    A = sp.csc_matrix((vals.ravel(), rows.ravel(), np.arange(0, (n + 1) * 9, 9)), shape=(m, n))
    return dict(A=A, cone=cone)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["socp", "mixed", "qp", "many_small"])
def test_device_equilibration_is_bit_identical_to_host(case):
    lib = capi.load("libscsamd.so")
    rng = np.random.default_rng(3)
    P = None
    if case == "socp":
        pr = problems.random_socp(3000, 9000, 12, seed=5)
    elif case == "mixed":
        pr = _mixed_cone_matrix(2)
    elif case == "many_small":
        pr = problems.random_socp(2000, 6000, 8, seed=9, q_fixed=8)
    else:
        pr = problems.random_socp(800, 2400, 10, seed=7)
        n = 800
        M = sp.random(n, n, density=0.01, random_state=4, format="csc")
        P = (M @ M.T + sp.diags(rng.uniform(0.1, 2.0, n))).tocsc()
    h = _equilibrate(lib, pr["A"], P, pr["cone"], 0)
    d = _equilibrate(lib, pr["A"], P, pr["cone"], 1)
    for a, b, name in zip(h, d, ("A.x", "P.x", "D", "E")):
        if a is None:
            assert b is None
            continue
        assert np.array_equal(a, b), (case, name, np.abs(a - b).max())
