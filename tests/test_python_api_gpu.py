"""scs_amd.solver.SCS: the scs-python-shaped object (reference docs/src/api/python.rst) over the C ABI.
One workspace across solves: warm start, update(b, c), QP with P, scalar cone sizes."""
import numpy as np
import pytest
import scipy.sparse as sp

from scs_amd import capi, problems
from scs_amd.solver import SCS, solve

pytestmark = pytest.mark.gpu


def test_solve_update_and_warm_start_on_one_workspace():
    pr = problems.random_socp(300, 900, 8, seed=2)
    data = dict(A=pr["A"], b=pr["b"], c=pr["c"])
    with SCS(data, pr["cone"], eps_abs=1e-7, eps_rel=1e-7, acceleration_lookback=0) as solver:
        r1 = solver.solve()
        assert r1["info"]["status"] == "solved" and r1["info"]["status_val"] == 1
        assert abs(r1["info"]["pobj"] - float(pr["c"] @ pr["x_opt"])) <= 1e-4 * max(1.0, abs(float(pr["c"] @ pr["x_opt"])))
        # same numbers as the plain C-ABI path
        lib = capi.load("libscsamd.so")
        prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
        r0 = capi.solve(lib, prob, verbose=0, eps_abs=1e-7, eps_rel=1e-7, acceleration_lookback=0)
        assert r0["info"]["iter"] == r1["info"]["iter"] and np.array_equal(r0["x"], r1["x"])
        # warm start from the solution: converges (almost) immediately
        r2 = solver.solve(warm_start=True)
        assert r2["info"]["status_val"] == 1 and r2["info"]["iter"] <= r1["info"]["iter"] // 4
        # cold again on request: the workspace keeps its adapted scale (as the reference's does,
        # src/scs.c:1236-1241 writes it into the settings), so the count differs, not the answer
        r3 = solver.solve(warm_start=False)
        assert r3["info"]["status_val"] == 1 and r3["info"]["iter"] > r2["info"]["iter"]
        np.testing.assert_allclose(r3["x"], r1["x"], rtol=0, atol=2e-4 * max(1.0, np.abs(r1["x"]).max()))
        # new right-hand side on the same workspace == a fresh solver on the modified data
        b2 = pr["b"].copy()
        z, l = pr["cone"]["z"], pr["cone"]["l"]
        b2[z:z + l] += 0.1                     # loosen the inequalities: still feasible and bounded
        solver.update(b=b2)
        r4 = solver.solve(warm_start=False)
        r5 = solve(dict(A=pr["A"], b=b2, c=pr["c"]), pr["cone"], eps_abs=1e-7, eps_rel=1e-7, acceleration_lookback=0)
        assert r4["info"]["status_val"] == r5["info"]["status_val"] == 1
        assert abs(r4["info"]["pobj"] - r5["info"]["pobj"]) <= 1e-5 * max(1.0, abs(r5["info"]["pobj"]))
        np.testing.assert_allclose(r4["x"], r5["x"], rtol=0, atol=2e-4 * max(1.0, np.abs(r5["x"]).max()))
    with pytest.raises(RuntimeError):
        solver.solve()


def test_qp_with_P_scalar_cone_sizes_and_legacy_zero_cone_name():
    rng = np.random.default_rng(0)
    n, m = 40, 60
    M = rng.standard_normal((n, n))
    P = sp.csc_matrix(M @ M.T / n + np.eye(n))
    A1, A3 = rng.standard_normal((10, n)), rng.standard_normal((10, n))
    A = sp.csc_matrix(np.vstack([A1, -np.eye(n), A3]))
    x0 = np.abs(rng.standard_normal(n))   # a feasible point: A1 x0 = b1, x0 >= 0, b3 - A3 x0 in SOC
    b3 = np.zeros(10)
    b3[0] = A3[0] @ x0 + np.linalg.norm(A3[1:] @ x0) + 1.0
    b = np.concatenate([A1 @ x0, np.zeros(n), b3])
    c = rng.standard_normal(n)
    cone = dict(f=10, l=n, q=10)            # legacy 'f', scalar 'q'
    r = solve(dict(P=P, A=A, b=b, c=c), cone, eps_abs=1e-8, eps_rel=1e-8)
    assert r["info"]["status_val"] == 1
    x, y, s = r["x"], r["y"], r["s"]
    # KKT of  min 1/2 x'Px + c'x  s.t. Ax + s = b, s in K
    assert np.abs(P @ x + c + A.T @ y).max() <= 1e-5 * max(1.0, np.abs(c).max())
    assert np.abs(A @ x + s - b).max() <= 1e-5 * max(1.0, np.abs(b).max())
    assert np.abs(s[:10]).max() <= 1e-6 and s[10:50].min() >= -1e-6 and s[50] >= np.linalg.norm(s[51:]) - 1e-6
    assert abs(float(y @ s)) <= 1e-5


def test_bad_inputs_raise_like_scs_python():
    pr = problems.random_socp(30, 90, 4, seed=1)
    with pytest.raises(ValueError):
        SCS(dict(A=pr["A"], b=pr["b"]), pr["cone"])                       # no 'c'
    with pytest.raises(ValueError):
        SCS(dict(A=pr["A"], b=pr["b"], c=pr["c"]), dict(l=5))              # cone rows != m
    with pytest.raises(AttributeError):
        SCS(dict(A=pr["A"], b=pr["b"], c=pr["c"]), pr["cone"], no_such_setting=1)
    s = SCS(dict(A=pr["A"], b=pr["b"], c=pr["c"]), pr["cone"], use_indirect=True, gpu=True)  # accepted, ignored
    with pytest.raises(ValueError):
        s.solve(x=np.zeros(3))
    s.close()
