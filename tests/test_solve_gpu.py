"""B2 parity: device-resident scs_solve vs the reference CPU indirect solver
(`oracle/_ref/libscsindir_ref.so`) on identical random_socp_prob-style inputs.
North-star bar: residuals / objectives within 1e-6 relative, same status."""
import numpy as np
import pytest

from scs_amd import capi, problems

pytestmark = pytest.mark.gpu

REL = 1e-6  # BASELINE.json north_star: "matching the CPU indirect solver to 1e-6 relative"


def _ref():
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built")
    return pyoracle.load_ref()


def _close(a, b, scale=1.0, rel=REL):
    return abs(a - b) <= rel * max(abs(a), abs(b), scale)


def _compare(ia, ir, x_a, x_r, pobj_scale=1.0):
    assert ia["status_val"] == ir["status_val"], (ia["status"], ir["status"])
    assert ia["iter"] == ir["iter"], (ia["iter"], ir["iter"])
    for k in ("pobj", "dobj"):
        assert _close(ia[k], ir[k], pobj_scale), (k, ia[k], ir[k])
    for k in ("res_pri", "res_dual", "gap"):
        # residuals are ~1e-4 * scale at termination: compare relative to the stopping scale
        assert abs(ia[k] - ir[k]) <= REL * max(1.0, pobj_scale), (k, ia[k], ir[k])
    assert np.abs(x_a - x_r).max() <= 1e-5 * max(1.0, np.abs(x_r).max())


@pytest.mark.parametrize("n,m,col_nnz,seed,over", [
    (200, 600, 8, 1, {}),
    (1000, 3000, 32, 1234, {}),                      # BASELINE config 1
    (1000, 3000, 32, 1234, dict(normalize=0)),
    (1000, 3000, 32, 7, dict(adaptive_scale=0, scale=1.0)),
    (3000, 9000, 10, 5, dict(eps_abs=1e-6, eps_rel=1e-6)),
])
def test_socp_matches_reference(n, m, col_nnz, seed, over):
    ref = _ref()
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(n, m, col_nnz, seed=seed)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, acceleration_lookback=0, **over)
    ra = capi.solve(amd, prob, **kw)
    rr = capi.solve(ref, prob, **kw)
    scale = max(1.0, abs(rr["info"]["pobj"]))
    _compare(ra["info"], rr["info"], ra["x"], rr["x"], scale)
    # and both are near the known optimum of the generator (problem_utils.h:22-81)
    popt = float(pr["c"] @ pr["x_opt"])
    assert abs(ra["info"]["pobj"] - popt) <= 5e-3 * max(1.0, abs(popt))


def test_many_small_socs_and_lp_only():
    ref = _ref()
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(500, 1500, 6, seed=3, q_fixed=5)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, acceleration_lookback=0)
    ra, rr = capi.solve(amd, prob, **kw), capi.solve(ref, prob, **kw)
    _compare(ra["info"], rr["info"], ra["x"], rr["x"], max(1.0, abs(rr["info"]["pobj"])))
    cone = dict(z=50, l=550)
    pr = problems.random_cone_prob(200, 600, 5, cone, seed=4)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    ra, rr = capi.solve(amd, prob, **kw), capi.solve(ref, prob, **kw)
    _compare(ra["info"], rr["info"], ra["x"], rr["x"], max(1.0, abs(rr["info"]["pobj"])))


def test_warm_start_and_update():
    import ctypes as C
    ref = _ref()
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(300, 900, 8, seed=11)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    out = {}
    for name, lib in (("amd", amd), ("ref", ref)):
        T = lib._scs_types
        st = capi.default_settings(lib, verbose=0, acceleration_lookback=0)
        x, y, s = np.zeros(prob.n), np.zeros(prob.m), np.zeros(prob.m)
        sol = T.ScsSolution(x.ctypes.data_as(T.fp), y.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp))
        info = T.ScsInfo()
        w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
        assert w
        lib.scs_solve(w, C.byref(sol), C.byref(info), 0)
        it_cold = info.iter
        lib.scs_solve(w, C.byref(sol), C.byref(info), 1)  # warm start from the solution
        it_warm = info.iter
        b2 = (pr["b"] * 1.01).copy()
        assert lib.scs_update(w, b2.ctypes.data_as(T.fp), None) == 0
        lib.scs_solve(w, C.byref(sol), C.byref(info), 1)
        out[name] = (it_cold, it_warm, info.iter, info.pobj, info.status_val)
        lib.scs_finish(w)
    assert out["amd"][:3] == out["ref"][:3], out
    assert out["amd"][4] == out["ref"][4]
    assert abs(out["amd"][3] - out["ref"][3]) <= 1e-6 * max(1.0, abs(out["ref"][3]))
    assert out["amd"][1] < out["amd"][0]
