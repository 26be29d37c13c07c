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
