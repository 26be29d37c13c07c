"""Randomised parity sweep: small conic programs over random MIXES of every cone type the
backend carries (zero, nonnegative, box, SOC, PSD, complex PSD, exponential, dual exponential,
power, dual power), feasible and bounded by the reference generator's construction
(test/problem_utils.h:43-56: z random, y = Proj_K*(z) with the REFERENCE's projection,
s = y - z, b = Ax + s, c = -A'y).  Each problem is solved by the reference build and by the HIP
library with default settings (Anderson acceleration on, inexact CG): same status, objectives
equal to the accuracy the termination criterion gives, and our solution passes the optimality
conditions computed independently here."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import pyoracle
from scs_amd import capi, problems

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")]
T = capi.T64


def _ref_proj_dual(ref, cone, x):
    k = capi.make_cone(cone)
    ref._scs_init_cone.restype = C.c_void_p
    ref._scs_init_cone.argtypes = [C.POINTER(T.ScsCone), C.c_int]
    ref._scs_proj_dual_cone.argtypes = [T.fp, C.c_void_p, C.c_void_p, T.fp]
    ref._scs_finish_cone.argtypes = [C.c_void_p]
    w = ref._scs_init_cone(C.byref(k), len(x))
    assert w
    out = np.array(x, dtype=np.float64)
    assert ref._scs_proj_dual_cone(out.ctypes.data_as(T.fp), w, None, None) == 0
    ref._scs_finish_cone(w)
    return out


def _random_cone(rng):
    cone = {}
    if rng.random() < 0.7:
        cone["z"] = int(rng.integers(1, 6))
    if rng.random() < 0.8:
        cone["l"] = int(rng.integers(1, 12))
    if rng.random() < 0.4:
        nb = int(rng.integers(1, 6))
        cone["bu"] = list(rng.uniform(0.5, 3.0, nb))
        cone["bl"] = list(-rng.uniform(0.5, 3.0, nb))
    if rng.random() < 0.8:
        cone["q"] = [int(v) for v in rng.integers(1, 20, size=rng.integers(1, 5))]
    if rng.random() < 0.6:
        cone["s"] = [int(v) for v in rng.integers(1, 9, size=rng.integers(1, 4))]
    if rng.random() < 0.3:
        cone["cs"] = [int(v) for v in rng.integers(1, 5, size=rng.integers(1, 3))]
    if rng.random() < 0.4:
        cone["ep"] = int(rng.integers(1, 4))
    if rng.random() < 0.4:
        cone["ed"] = int(rng.integers(1, 4))
    if rng.random() < 0.4:
        cone["p"] = [float(v) for v in rng.uniform(-0.9, 0.9, size=rng.integers(1, 4))]
    if not cone:
        cone["l"] = 5
    return cone


@pytest.mark.parametrize("renumber", [False, True])
@pytest.mark.parametrize("seed", range(16))
def test_random_mixed_cone_program_matches_reference(seed, renumber, monkeypatch):
    """renumber = True (round 6): the same sweep with scs_init's internal numbering FORCED (option reorder = 1: at these sizes the library
    would not bother) -- variables, zero / nonnegative rows and the tails of the second-order cones are permuted inside, box / PSD /
    exponential / power rows must come through untouched, and x, y, s are handed back in the caller's order."""
    if renumber:
        monkeypatch.setenv("SCS_AMD_REORDER", "1")
    ref = pyoracle.load_ref()
    amd = capi.load("libscsamd.so")
    rng = np.random.default_rng(1000 + seed)
    cone = _random_cone(rng)
    m = capi.cone_rows(cone)
    n = max(2, m // int(rng.integers(2, 5)))
    z = rng.standard_normal(m)
    y = _ref_proj_dual(ref, cone, z)
    s = y - z
    x = rng.standard_normal(n)
    A = sp.random(m, n, density=min(1.0, 6.0 / n), random_state=seed, format="csc", data_rvs=rng.standard_normal)
    A = (A + sp.csc_matrix((np.full(min(m, n), 0.5), (np.arange(min(m, n)), np.arange(min(m, n)))), shape=(m, n))).tocsc()
    b = A @ x + s
    c = -(A.T @ y)
    prob = capi.Problem(A, b, c, cone)
    kw = dict(verbose=0, eps_abs=1e-7, eps_rel=1e-7, max_iters=20000)
    ra = capi.solve(amd, prob, **kw)
    rr = capi.solve(ref, prob, **kw)
    ia, ir = ra["info"], rr["info"]
    assert ia["status_val"] == ir["status_val"] == 1, (cone, ia["status"], ir["status"])
    scale = max(1.0, abs(ir["pobj"]))
    assert abs(ia["pobj"] - ir["pobj"]) <= 2e-5 * scale, (cone, ia["pobj"], ir["pobj"])
    assert abs(ia["pobj"] - ia["dobj"]) <= 2e-5 * scale
    # the reference generator's certificate: (x, y, s) built above is optimal, so c'x is THE optimum
    assert abs(ia["pobj"] - float(c @ x)) <= 5e-5 * max(1.0, abs(float(c @ x))), cone
    # independent optimality check of OUR solution: Ax + s = b, A'y + c = 0, y in K* and s in K up to eps
    xa, ya, sa = ra["x"], ra["y"], ra["s"]
    assert np.abs(A @ xa + sa - b).max() <= 1e-5 * max(1.0, np.abs(b).max())
    assert np.abs(A.T @ ya + c).max() <= 1e-5 * max(1.0, np.abs(c).max())
    assert np.abs(_ref_proj_dual(ref, cone, ya) - ya).max() <= 1e-6 * max(1.0, np.abs(ya).max())   # y in K*
    assert np.abs(_ref_proj_dual(ref, cone, -sa)).max() <= 1e-6 * max(1.0, np.abs(sa).max())       # s in K
    assert abs(float(ya @ sa)) <= 1e-4 * max(1.0, np.abs(ya).max() * np.abs(sa).max())


def _base_problem(seed):
    rng = np.random.default_rng(5000 + seed)
    cone = dict(z=2, l=int(rng.integers(4, 10)), q=[int(v) for v in rng.integers(2, 8, size=2)], s=[3])
    m = capi.cone_rows(cone)
    n = max(3, m // 3)
    A = sp.random(m, n, density=min(1.0, 5.0 / n), random_state=seed, format="lil", data_rvs=rng.standard_normal)
    return rng, cone, m, n, A


@pytest.mark.parametrize("renumber", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_infeasible_programs_get_the_reference_status_and_a_certificate(seed, renumber, monkeypatch):
    """Two contradictory rows in the nonnegative cone (x0 <= -1 and x0 >= 1): primal infeasible.
    Same status as the reference (SCS_INFEASIBLE = -2) and a valid certificate: A'y ~ 0, b'y < 0, y in K*.
    (renumber: the certificate comes back through the internal numbering, forced.)"""
    if renumber:
        monkeypatch.setenv("SCS_AMD_REORDER", "1")
    ref = pyoracle.load_ref()
    amd = capi.load("libscsamd.so")
    rng, cone, m, n, A = _base_problem(seed)
    b = rng.standard_normal(m)
    l0 = cone["z"]                       # first two rows of the l-block
    A[l0, :] = 0; A[l0, 0] = 1.0; b[l0] = -1.0       #  x0 + s = -1, s >= 0  ->  x0 <= -1
    A[l0 + 1, :] = 0; A[l0 + 1, 0] = -1.0; b[l0 + 1] = -1.0   # -x0 + s = -1       ->  x0 >= 1
    A = sp.csc_matrix(A)
    c = rng.standard_normal(n)
    prob = capi.Problem(A, b, c, cone)
    ra = capi.solve(amd, prob, verbose=0)
    rr = capi.solve(ref, prob, verbose=0)
    assert ra["info"]["status_val"] == rr["info"]["status_val"] == -2, (ra["info"]["status"], rr["info"]["status"])
    y = ra["y"]
    assert float(b @ y) < 0
    y = y / -float(b @ y)                # normalised like the reference: b'y = -1
    assert np.abs(A.T @ y).max() <= 1e-5
    assert np.abs(_ref_proj_dual(ref, cone, y) - y).max() <= 1e-6 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("seed", range(6))
def test_unbounded_programs_get_the_reference_status_and_a_certificate(seed):
    """A variable that no row touches, with negative cost: unbounded below.  Same status as the
    reference (SCS_UNBOUNDED = -1) and a valid certificate: Ax + s ~ 0, s in K, c'x < 0."""
    ref = pyoracle.load_ref()
    amd = capi.load("libscsamd.so")
    rng, cone, m, n, A = _base_problem(seed + 50)
    A[:, n - 1] = 0                      # last variable is free of every constraint ...
    A = sp.csc_matrix(A)
    c = np.abs(rng.standard_normal(n))
    c[n - 1] = -1.0                      # ... and pays for going to +infinity
    b = np.abs(rng.standard_normal(m)) + 0.5
    prob = capi.Problem(A, b, c, cone)
    ra = capi.solve(amd, prob, verbose=0)
    rr = capi.solve(ref, prob, verbose=0)
    assert ra["info"]["status_val"] == rr["info"]["status_val"] == -1, (ra["info"]["status"], rr["info"]["status"])
    x, s = ra["x"], ra["s"]
    cx = float(c @ x)
    assert cx < 0
    x, s = x / -cx, s / -cx              # c'x = -1
    assert np.abs(A @ x + s).max() <= 1e-5
    assert np.abs(_ref_proj_dual(ref, cone, -s)).max() <= 1e-6 * max(1.0, np.abs(s).max())   # s in K
