"""CPU tests (no GPU): the plain-C restatement oracle/scs_oracle.c is pinned against
the golden vectors captured from the real reference (tests/golden/) and, where
oracle/_ref/ is present, against the reference itself."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import pyoracle
from scs_amd import capi, problems

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_linsys_restatement_vs_golden_boundary_vectors():
    g = np.load(os.path.join(G, "linsys_cfg1.npz"))
    n, m = int(g["n"]), int(g["m"])
    ls = pyoracle.OracleLinSys(m, n, g["Ap"], g["Ai"], g["Ax_normalized"], g["diag_r"])
    A = sp.csc_matrix((g["Ax_normalized"], g["Ai"], g["Ap"]), shape=(m, n))
    rx, ry = g["diag_r"][:n], g["diag_r"][n:]
    for c in g["calls"]:
        tol = float(g[f"tol{c}"])
        out, its = ls.solve(g[f"b{c}"], g[f"s{c}"], tol)
        want = g[f"xy{c}"]
        err = np.abs(out - want).max() / np.abs(want).max()
        rhs = g[f"b{c}"]
        red = rx * out[:n] + A.T @ ((A @ out[:n]) / ry) - (rhs[:n] + A.T @ (rhs[n:] / ry))
        assert np.abs(red).max() < max(tol, 1e-12) * (1 + 1e-9) + 1e-13 * np.abs(rhs).max()
        if tol <= 1e-9:
            assert err <= 1e-9, (int(c), err)
        else:
            assert err <= 50 * tol, (int(c), tol, err)  # CG at loose tol: O(tol) spread, see DESIGN.md
    ls.close()


def test_cone_restatement_vs_golden():
    g = np.load(os.path.join(G, "cones.npz"))
    meta = json.load(open(os.path.join(G, "cones_meta.json")))
    seen = set()
    for name, cone in meta.items():
        seen.update(k for k in ("ep", "ed", "p", "cs", "s", "q", "bu") if cone.get(k))
        for variant in ("eucl", "ry"):
            x = g[f"{name}_{variant}_x"]
            want = g[f"{name}_{variant}_y"]
            r = g[f"{name}_{variant}_r"] if variant == "ry" else None
            got = pyoracle.oracle_proj_dual_cone(cone, x, r)
            err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
            tol = 1e-11 if ("psd" in name or name in ("mixed", "all", "all_c")) else 1e-12
            assert err <= tol, (name, variant, err)
    # the restatement covers every cone the fixtures hold: SOC, box, PSD, complex PSD, exp, dual exp, power
    assert {"q", "bu", "s", "cs", "ep", "ed", "p"} <= seen, seen


def _golden_solves():
    return json.load(open(os.path.join(G, "solves.json")))


def _problem(spec):
    if spec["kind"] == "socp":
        pr = problems.random_socp(spec["n"], spec["m"], spec["col_nnz"], seed=spec["seed"], q_fixed=spec.get("q_fixed"))
    else:
        pr = problems.random_sdp(spec["n"], spec["n_blocks"], spec["block"], spec["bsize"], spec["col_nnz"],
                                 seed=spec["seed"])
    return pr, capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])


def test_generator_reproduces_golden_inputs():
    for rec in _golden_solves():
        pr, prob = _problem(rec["spec"])
        assert abs(float(np.abs(prob.Ax).sum() + np.abs(prob.b).sum()) - rec["data_checksum"]) <= 1e-9 * rec["data_checksum"]


@pytest.mark.parametrize("idx", [0, 4, 5])
def test_whole_solve_restatement_vs_golden_info(idx):
    rec = _golden_solves()[idx]
    assert rec["spec"]["over"].get("acceleration_lookback") == 0
    pr, prob = _problem(rec["spec"])
    over = {k: v for k, v in rec["spec"]["over"].items() if k != "acceleration_lookback"}
    r = pyoracle.oracle_solve(prob, **over)
    gi = rec["info"]
    assert r["info"]["status_val"] == gi["status_val"] == 1
    scale = max(1.0, abs(gi["pobj"]))
    assert abs(r["info"]["pobj"] - gi["pobj"]) <= 1e-3 * scale
    assert abs(r["info"]["dobj"] - gi["dobj"]) <= 1e-3 * scale
    assert 0.5 <= r["info"]["iter"] / gi["iter"] <= 2.0


@pytest.mark.skipif(not pyoracle.ref_available("libscsindir_ref_exactcg.so"), reason="oracle/_ref not built")
@pytest.mark.parametrize("n,m,cn,seed,over", [(120, 360, 6, 1, {}), (200, 600, 8, 2, dict(normalize=0)),
                                              (150, 500, 5, 3, dict(adaptive_scale=0))])
def test_restatement_trajectory_equals_reference_with_exact_cg(n, m, cn, seed, over):
    """ADMM map restated faithfully: with both sides solving to the 1e-12 floor the
    runs coincide (iteration count, scale updates, residuals to 1e-6 relative)."""
    ref = pyoracle.load_ref("libscsindir_ref_exactcg.so")
    pr = problems.random_socp(n, m, cn, seed=seed)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    rr = capi.solve(ref, prob, verbose=0, acceleration_lookback=0, **over)
    ro = pyoracle.oracle_solve(prob, cg_tol_override=1e-12, **over)
    assert ro["info"]["iter"] == rr["info"]["iter"]
    assert ro["info"]["scale_updates"] == rr["info"]["scale_updates"]
    for k in ("pobj", "dobj", "res_pri", "res_dual", "gap", "scale"):
        assert abs(ro["info"][k] - rr["info"][k]) <= 1e-6 * max(abs(rr["info"][k]), 1e-3), k
    assert np.abs(ro["x"] - rr["x"]).max() <= 1e-6 * max(1.0, np.abs(rr["x"]).max())


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
def test_sdp_box_restatement_vs_reference_exact_cg():
    ref = pyoracle.load_ref("libscsindir_ref_exactcg.so")
    pr = problems.random_sdp(40, n_blocks=4, block=6, bsize=11, col_nnz=5, seed=9)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    rr = capi.solve(ref, prob, verbose=0, acceleration_lookback=0)
    ro = pyoracle.oracle_solve(prob, cg_tol_override=1e-12)
    assert ro["info"]["iter"] == rr["info"]["iter"]
    assert abs(ro["info"]["pobj"] - rr["info"]["pobj"]) <= 1e-6 * max(1.0, abs(rr["info"]["pobj"]))


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
def test_spmv_restatement_bitwise_vs_reference_kernel():
    import ctypes as C
    ref = pyoracle.load_ref()
    lib = pyoracle.restatement()
    rng = np.random.default_rng(0)
    from tests import probgen
    A = probgen.random_csc(300, 100, 7, seed=3)
    prob = capi.Problem(A, np.zeros(300), np.zeros(100), dict(l=300))
    x = rng.uniform(-1, 1, 300)
    y0 = rng.uniform(-1, 1, 100)
    y1, y2 = y0.copy(), y0.copy()
    T = capi.T64
    ref._scs_accum_by_atrans(C.byref(prob.matA), x.ctypes.data_as(T.fp), y1.ctypes.data_as(T.fp))
    lib.or_accum_by_atrans(100, prob.Ap.ctypes.data_as(T.ip), prob.Ai.ctypes.data_as(T.ip), prob.Ax.ctypes.data_as(T.fp),
                           x.ctypes.data_as(T.fp), y2.ctypes.data_as(T.fp))
    assert np.abs(y1 - y2).max() <= 4e-16 * np.abs(y1).max()


@pytest.mark.skipif(not pyoracle.ref_available("libscsindir_ref_exactcg.so"), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_all_cone_restatement_trajectory_equals_reference_with_exact_cg(seed):
    """The whole-solve restatement over EVERY cone type (box, SOC, PSD, complex PSD, exponential,
    dual exponential, power, dual power) walks the reference's trajectory when both solve the
    linear systems to the 1e-12 floor.  The program is feasible and bounded by the reference
    generator's construction, with the reference's own projection (test/problem_utils.h:43-56)."""
    import ctypes as C
    import scipy.sparse as sp
    ref = pyoracle.load_ref("libscsindir_ref_exactcg.so")
    T = capi.T64
    rng = np.random.default_rng(300 + seed)
    cone = dict(z=3, l=6, bu=[1.0, 2.0], bl=[-1.0, -0.5], q=[4, 7], s=[3, 5], cs=[3], ep=2, ed=2, p=[0.4, -0.7])
    m = capi.cone_rows(cone)
    n = m // 3
    k = capi.make_cone(cone)
    ref._scs_init_cone.restype = C.c_void_p
    ref._scs_init_cone.argtypes = [C.POINTER(T.ScsCone), C.c_int]
    ref._scs_proj_dual_cone.argtypes = [T.fp, C.c_void_p, C.c_void_p, T.fp]
    ref._scs_finish_cone.argtypes = [C.c_void_p]
    z = rng.standard_normal(m)
    y = z.copy()
    w = ref._scs_init_cone(C.byref(k), m)
    assert ref._scs_proj_dual_cone(y.ctypes.data_as(T.fp), w, None, None) == 0
    ref._scs_finish_cone(w)
    # the restatement's projection agrees with the reference's on this very vector
    assert np.abs(pyoracle.oracle_proj_dual_cone(cone, z) - y).max() <= 1e-10 * max(1.0, np.abs(y).max())
    s = y - z
    x = rng.standard_normal(n)
    A = sp.random(m, n, density=min(1.0, 5.0 / n), random_state=seed, format="csc", data_rvs=rng.standard_normal)
    A = (A + sp.csc_matrix((np.full(n, 0.7), (np.arange(n), np.arange(n))), shape=(m, n))).tocsc()
    prob = capi.Problem(A, A @ x + s, -(A.T @ y), cone)
    kw = dict(eps_abs=1e-5, eps_rel=1e-5, max_iters=30000)
    rr = capi.solve(ref, prob, verbose=0, acceleration_lookback=0, **kw)
    ro = pyoracle.oracle_solve(prob, cg_tol_override=1e-12, **kw)
    assert rr["info"]["status_val"] == ro["info"]["status_val"] == 1
    assert ro["info"]["iter"] == rr["info"]["iter"]
    assert ro["info"]["scale_updates"] == rr["info"]["scale_updates"]
    for key in ("pobj", "dobj", "res_pri", "res_dual", "gap", "scale"):
        assert abs(ro["info"][key] - rr["info"][key]) <= 1e-6 * max(abs(rr["info"][key]), 1e-3), key
    assert np.abs(ro["x"] - rr["x"]).max() <= 1e-6 * max(1.0, np.abs(rr["x"]).max())
