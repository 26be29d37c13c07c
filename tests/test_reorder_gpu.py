"""Locality by construction end to end (scs_amd/csrc/reorder.h): a problem whose variables and zero / nonnegative rows arrive in an
arbitrary numbering is renumbered inside scs_init, solved in the new numbering, and handed back in the caller's -- the caller
sees the same answer as for the un-scrambled problem.  VERDICT r3 item 4: "solutions equal to the un-permuted problem's to 1e-9"."""
import ctypes as C

import numpy as np
import pytest

from scs_amd import capi, problems

pytestmark = pytest.mark.gpu
N, M, BAND = 30000, 60000, 512


def _solve(lib, pr, warm=None, **kw):
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    return capi.solve(lib, prob, verbose=0, acceleration_lookback=0, warm=warm, **kw), prob


def _reorder_info(lib, prob):
    st = capi.default_settings(lib, verbose=0)
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    assert w
    out = (C.c_double * 6)()
    lib.scs_amd_get_reorder_info(w, out)
    lib.scs_finish(w)
    return list(out)


def test_scrambled_problem_gives_the_unscrambled_answer_exact_cg(monkeypatch):
    lib = capi.load("libscsamd.so")
    plain = problems.random_socp(N, M, 10, seed=21, band=BAND)
    scr = problems.scramble_prob(plain, 6)
    monkeypatch.setenv("SCS_AMD_REORDER", "0")
    r0, _ = _solve(lib, plain, cg_tol_override=1e-12, max_iters=150)          # the problem as generated, no renumbering
    rs_off, _ = _solve(lib, scr, cg_tol_override=1e-12, max_iters=150)        # scrambled, solved as given
    monkeypatch.setenv("SCS_AMD_REORDER", "1")                                # (forced: 3e5 nonzeros is below the library's threshold)
    rs_on, prob = _solve(lib, scr, cg_tol_override=1e-12, max_iters=150)      # scrambled, renumbered inside scs_init
    info = _reorder_info(lib, prob)
    assert info[0] == 1.0 and 0.5 * (info[3] + info[4]) < 0.5 * 0.5 * (info[1] + info[2]), info
    cp, rp = scr["col_perm"], scr["row_perm"]
    for r in (rs_off, rs_on):
        assert r["info"]["iter"] == r0["info"]["iter"] == 150
        for v, perm in (("x", cp), ("y", rp), ("s", rp)):
            want = r0[v][perm]
            d = np.abs(r[v] - want).max() / max(1.0, np.abs(want).max())
            assert d <= 1e-9, (v, d)
        for k in ("pobj", "dobj", "res_pri", "res_dual", "gap"):
            assert abs(r["info"][k] - r0["info"][k]) <= 1e-8 * max(1.0, abs(r0["info"][k])), k


def test_default_schedule_to_termination_warm_start_and_update_through_the_renumbering(monkeypatch):
    lib = capi.load("libscsamd.so")
    scr = problems.random_socp(N, M, 10, seed=22, band=BAND, scramble=3)
    monkeypatch.setenv("SCS_AMD_REORDER", "0")
    off, _ = _solve(lib, scr)
    monkeypatch.setenv("SCS_AMD_REORDER", "1")
    on, prob = _solve(lib, scr)
    assert on["info"]["status_val"] == off["info"]["status_val"] == 1
    scale = max(1.0, abs(off["info"]["pobj"]))
    assert abs(on["info"]["pobj"] - off["info"]["pobj"]) <= 1e-3 * scale
    # the returned point is in the CALLER's numbering: residual identities recomputed from the caller's data
    A = prob.sparse()
    assert abs(np.abs(A @ on["x"] + on["s"] - prob.b).max() - on["info"]["res_pri"]) <= 1e-9 * max(1, on["info"]["res_pri"] * 1e4)
    assert abs(np.abs(A.T @ on["y"] + prob.c).max() - on["info"]["res_dual"]) <= 1e-9 * max(1, on["info"]["res_dual"] * 1e4)
    popt = float(scr["c"] @ scr["x_opt"])
    assert abs(on["info"]["pobj"] - popt) <= 5e-3 * max(1.0, abs(popt))
    # warm start from the returned point (mapped into the internal numbering on the way in): converged at the first check
    ws, _ = _solve(lib, scr, warm=(on["x"], on["y"], on["s"]))
    assert ws["info"]["status_val"] == 1 and ws["info"]["iter"] <= 50, ws["info"]["iter"]
    # scs_update with new b, c in the caller's numbering == a fresh scs_init on the updated problem
    T = lib._scs_types
    st = capi.default_settings(lib, verbose=0, acceleration_lookback=0, max_iters=100)
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    assert w
    b2, c2 = (prob.b * 1.05).copy(), (prob.c * 0.9).copy()
    assert lib.scs_update(w, b2.ctypes.data_as(T.fp), c2.ctypes.data_as(T.fp)) == 0
    x, y, s = np.zeros(prob.n), np.zeros(prob.m), np.zeros(prob.m)
    sol = T.ScsSolution(x.ctypes.data_as(T.fp), y.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp))
    inf = T.ScsInfo()
    lib.scs_solve(w, C.byref(sol), C.byref(inf), 0)
    lib.scs_finish(w)
    fresh = capi.solve(lib, capi.Problem(scr["A"], b2, c2, scr["cone"]), verbose=0, acceleration_lookback=0, max_iters=100)
    assert inf.iter == fresh["info"]["iter"] and np.array_equal(x, fresh["x"]) and np.array_equal(s, fresh["s"])


def test_pure_lp_renumbered_through_the_graph_search_gives_the_same_answer(monkeypatch):
    """no row that cannot move (zero + nonnegative cones only): scs_init renumbers both the variables and ALL rows (Cuthill-McKee);
    same answer as with the renumbering off, in the caller's numbering"""
    lib = capi.load("libscsamd.so")
    n, m = 20000, 50000
    cone = dict(z=15000, l=35000, q=[])
    scr = problems.scramble_prob(problems.random_cone_prob(n, m, 8, cone, seed=31, band=400), 2)
    monkeypatch.setenv("SCS_AMD_REORDER", "0")
    off, _ = _solve(lib, scr, cg_tol_override=1e-12, max_iters=120)
    monkeypatch.setenv("SCS_AMD_REORDER", "1")
    on, prob = _solve(lib, scr, cg_tol_override=1e-12, max_iters=120)
    assert _reorder_info(lib, prob)[0] == 1.0
    for v in ("x", "y", "s"):
        d = np.abs(on[v] - off[v]).max() / max(1.0, np.abs(off[v]).max())
        assert d <= 1e-8, (v, d)   # measured 1.2e-9: 120 ADMM iterations of 1e-12 solves in two summation orders
    for k in ("pobj", "dobj", "res_pri", "res_dual", "gap"):
        assert abs(on["info"][k] - off["info"][k]) <= 1e-7 * max(1.0, abs(off["info"][k])), k


def _session(lib, prob, b2, c2, warm0, exact, **over):
    """scs_init -> cold solve -> scs_update(b2, c2) -> warm solve from the previous solution -> (fresh workspace) solve warm-started from
    `warm0`, all through the public API of `lib` (ours or the reference's); returns the three (x, y, s, info) records."""
    T = lib._scs_types
    st = capi.default_settings(lib, verbose=0, acceleration_lookback=0, **over)
    out = []
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    assert w
    if exact:
        lib.scs_amd_set_cg_tol_override(w, 1e-12)
    x, y, s = np.zeros(prob.n), np.zeros(prob.m), np.zeros(prob.m)
    sol = T.ScsSolution(x.ctypes.data_as(T.fp), y.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp))
    inf = T.ScsInfo()
    lib.scs_solve(w, C.byref(sol), C.byref(inf), 0)
    out.append(dict(x=x.copy(), y=y.copy(), s=s.copy(), info=capi.info_dict(inf)))
    assert lib.scs_update(w, b2.ctypes.data_as(T.fp), c2.ctypes.data_as(T.fp)) == 0
    lib.scs_solve(w, C.byref(sol), C.byref(inf), 1)  # warm: the caller's vectors go IN through the renumbering too
    out.append(dict(x=x.copy(), y=y.copy(), s=s.copy(), info=capi.info_dict(inf)))
    lib.scs_finish(w)
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    assert w
    if exact:
        lib.scs_amd_set_cg_tol_override(w, 1e-12)
    x[:], y[:], s[:] = warm0
    lib.scs_solve(w, C.byref(sol), C.byref(inf), 1)
    out.append(dict(x=x.copy(), y=y.copy(), s=s.copy(), info=capi.info_dict(inf)))
    lib.scs_finish(w)
    return out


@pytest.mark.parametrize("family", ["scrambled_band", "uniformly_random"])
def test_renumbered_solve_warm_start_and_update_match_the_reference_exact_cg(monkeypatch, family):
    """the renumbering maps b, c, warm starts, scs_update vectors and (x, y, s) across the API
    boundary; the tests above compare the library with itself.  Here the REFERENCE (oracle/_ref exactcg flavour: every linear
    system to the 1e-12 floor, include/glbopts.h:253-255 hook) solves a scrambled banded SOCP with mixed zero / nonnegative /
    second-order cones in the caller's numbering, and this library solves it with the renumbering ON: equal iteration counts and
    1e-6 on every ScsInfo figure and on x, y, s in the CALLER's order -- for the cold solve, for the solve after scs_update(b, c),
    and for a solve warm-started from a perturbed point (cone row order is an ABI fact, include/scs.h:121-172)."""
    from oracle import pyoracle
    if not pyoracle.ref_available("libscsindir_ref_exactcg.so"):
        pytest.skip("oracle/_ref/libscsindir_ref_exactcg.so not built")
    ref = pyoracle.load_ref("libscsindir_ref_exactcg.so")
    lib = capi.load("libscsamd.so")
    n, m = (40000, 80000) if family == "scrambled_band" else (8000, 16000)  # (the reference's exact-CG legs take minutes on one host core)
    if family == "scrambled_band":   # hidden locality: recovered by the anchors / Cuthill-McKee numberings of round 4
        scr = problems.scramble_prob(problems.random_socp(n, m, 10, seed=23, band=BAND), 5)
    else:                            # round 6: no locality to recover -> chain + home numbering, which also PERMUTES THE TAILS of the
        scr = problems.random_socp(n, m, 10, seed=24)  # second-order cones (every row but the cone's first): y and s must come back in place
    assert scr["cone"]["z"] > 0 and scr["cone"]["l"] > 0 and len(scr["cone"]["q"]) > 2
    prob = capi.Problem(scr["A"], scr["b"], scr["c"], scr["cone"])
    rng = np.random.default_rng(5)
    b2 = (prob.b * 1.05).copy()  # uniform: stays feasible
    c2 = (prob.c * 0.9).copy()
    warm0 = (0.1 * rng.standard_normal(prob.n), 0.1 * rng.standard_normal(prob.m), 0.1 * rng.standard_normal(prob.m))
    over = dict(eps_abs=1e-2, eps_rel=1e-2, max_iters=400)  # exact CG on one host core: ~0.3 s per ADMM iteration at this size
    want = _session(ref, prob, b2, c2, warm0, exact=False, **over)
    monkeypatch.setenv("SCS_AMD_REORDER", "1")  # forced: 4e5 nonzeros is below the library's own threshold
    info = _reorder_info(lib, prob)
    assert info[0] == 1.0, info
    if family == "scrambled_band":
        assert 0.5 * (info[3] + info[4]) < 0.25 * (info[1] + info[2]), info
    else:
        assert 0.5 * (info[3] + info[4]) < 0.95 * 0.5 * (info[1] + info[2]), info   # (a small vector shares lines by chance already: 0.43 / 0.63 as given)
        T = lib._scs_types
        cp, rp = np.zeros(prob.n, dtype=T.np_int), np.zeros(prob.m, dtype=T.np_int)
        assert lib.scs_amd_plan_reorder(C.byref(prob.matA), C.byref(prob.k), cp.ctypes.data_as(T.ip), rp.ctypes.data_as(T.ip), None) == 1
        o = scr["cone"]["z"] + scr["cone"]["l"]
        assert np.count_nonzero(rp[o:] != np.arange(o, prob.m)) > 0.5 * (prob.m - o)   # the SOC tails really moved
    got = _session(lib, prob, b2, c2, warm0, exact=True, **over)
    for stage, (g, r) in enumerate(zip(got, want)):
        gi, ri = g["info"], r["info"]
        assert gi["status_val"] == ri["status_val"] == 1, (stage, gi["status"], ri["status"])
        assert gi["iter"] == ri["iter"], (stage, gi["iter"], ri["iter"])
        assert gi["scale_updates"] == ri["scale_updates"], stage
        for k in ("pobj", "dobj", "res_pri", "res_dual", "gap", "scale"):
            assert abs(gi[k] - ri[k]) <= 1e-6 * max(abs(gi[k]), abs(ri[k]), 1e-3), (stage, k, gi[k], ri[k])
        for v in ("x", "y", "s"):
            d = np.abs(g[v] - r[v]).max() / max(1.0, np.abs(r[v]).max())
            assert d <= 1e-6, (stage, v, d)
    assert want[1]["info"]["iter"] < want[0]["info"]["iter"]  # the warm solve after the update really started from the old point
