"""B2 parity: device-resident scs_solve vs the reference CPU indirect solver on
identical random_socp_prob-style inputs.

Two tiers (DESIGN.md "Parity"):
 * exact-CG trajectory parity -- both sides solve every linear system to the 1e-12
   floor (reference flavour `exactcg`, built from the unmodified sources with the
   documented CG_NORM override; ours via scs_amd_set_cg_tol_override).  Then the
   ADMM map is evaluated to rounding on both sides and the runs must agree in
   iteration count, status and every ScsInfo residual/objective to 1e-6 relative
   (the north-star bar).
 * default inexact-CG schedule -- CG iterates at loose tolerances depend on the
   summation order to O(tol) (the reference disagrees with an instruction-for-
   instruction numpy restatement of itself by 3e-4 after 94 CG steps), so
   trajectories legitimately differ; both must reach the same status and optimum.
"""
import ctypes as C

import numpy as np
import pytest

from scs_amd import capi, problems

pytestmark = pytest.mark.gpu
REL = 1e-6


def _ref(name="libscsindir_ref.so"):
    from oracle import pyoracle
    if not pyoracle.ref_available(name):
        pytest.skip(f"oracle/_ref/{name} not built")
    return pyoracle.load_ref(name)


def _rel(a, b, floor=1.0):
    return abs(a - b) / max(abs(a), abs(b), floor)


CASES = [
    (200, 600, 8, 1, {}, None),
    (1000, 3000, 32, 1234, {}, None),                       # BASELINE configs[0]
    (1000, 3000, 32, 1234, dict(normalize=0), None),
    (1000, 3000, 32, 7, dict(adaptive_scale=0, scale=1.0), None),
    (500, 1500, 6, 3, {}, 5),                               # many small SOCs
    (800, 2000, 10, 5, dict(eps_abs=1e-7, eps_rel=1e-7), None),
]


@pytest.mark.parametrize("n,m,col_nnz,seed,over,q_fixed", CASES)
def test_exact_cg_trajectory_parity(n, m, col_nnz, seed, over, q_fixed):
    ref = _ref("libscsindir_ref_exactcg.so")
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(n, m, col_nnz, seed=seed, q_fixed=q_fixed)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, acceleration_lookback=0, **over)
    ra = capi.solve(amd, prob, cg_tol_override=1e-12, **kw)
    rr = capi.solve(ref, prob, **kw)
    ia, ir = ra["info"], rr["info"]
    assert ia["status_val"] == ir["status_val"] == 1
    assert ia["iter"] == ir["iter"], (ia["iter"], ir["iter"])
    assert ia["scale_updates"] == ir["scale_updates"]
    for k in ("pobj", "dobj", "res_pri", "res_dual", "gap", "scale"):
        assert _rel(ia[k], ir[k], floor=1e-3) <= REL, (k, ia[k], ir[k])
    for v in ("x", "y", "s"):
        d = np.abs(ra[v] - rr[v]).max() / max(1.0, np.abs(rr[v]).max())
        assert d <= REL, (v, d)


@pytest.mark.parametrize("n,m,col_nnz,seed,over,q_fixed", [CASES[1], CASES[4]])
def test_exact_cg_trajectory_parity_through_the_three_kernel_iteration(monkeypatch, n, m, col_nnz, seed, over, q_fixed):
    """the same 1e-6 trajectory bar with the PCG loop of the large systems forced on at a size the reference solves in seconds: wave-owned-
    rows products (SCS_AMD_WAVEROWS=1), no graph replay, k_cg3_update (SCS_AMD_CG3=1: stop test, alpha, beta and the vector updates in one
    launch, beta from the expanded z'r)"""
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")
    monkeypatch.setenv("SCS_AMD_GRAPH", "0")
    monkeypatch.setenv("SCS_AMD_CG3", "1")
    test_exact_cg_trajectory_parity(n, m, col_nnz, seed, over, q_fixed)


@pytest.mark.parametrize("n,m,col_nnz,seed,over,q_fixed", CASES[:5])
def test_default_schedule_same_optimum(n, m, col_nnz, seed, over, q_fixed):
    ref = _ref()
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(n, m, col_nnz, seed=seed, q_fixed=q_fixed)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, acceleration_lookback=0, **over)
    ra, rr = capi.solve(amd, prob, **kw), capi.solve(ref, prob, **kw)
    ia, ir = ra["info"], rr["info"]
    assert ia["status_val"] == ir["status_val"] == 1
    assert 0.5 <= ia["iter"] / ir["iter"] <= 2.0, (ia["iter"], ir["iter"])
    scale = max(1.0, abs(ir["pobj"]))
    # both stopped on eps = 1e-4 criteria: objectives agree to a few eps
    assert abs(ia["pobj"] - ir["pobj"]) <= 1e-3 * scale
    assert abs(ia["dobj"] - ir["dobj"]) <= 1e-3 * scale
    # independent check of OUR answer in the manner of test/problem_utils.h:107-249
    A = prob.sparse()
    x, y, s = ra["x"], ra["y"], ra["s"]
    res_pri = np.abs(A @ x + s - prob.b).max()
    res_dual = np.abs(A.T @ y + prob.c).max()
    assert abs(res_pri - ia["res_pri"]) <= 1e-9 * max(1, res_pri * 1e4)
    assert abs(res_dual - ia["res_dual"]) <= 1e-9 * max(1, res_dual * 1e4)
    eps = 1e-4
    assert res_pri <= eps + eps * max(np.abs(prob.b).max(), np.abs(s).max(), np.abs(A @ x).max())
    assert res_dual <= eps + eps * max(np.abs(prob.c).max(), np.abs(A.T @ y).max())
    assert abs(prob.c @ x + prob.b @ y) <= eps + eps * max(abs(prob.c @ x), abs(prob.b @ y))
    assert abs(s @ y) <= 5e-8 * max(np.abs(s).max(), np.abs(y).max()) * 10
    popt = float(pr["c"] @ pr["x_opt"])
    assert abs(ia["pobj"] - popt) <= 5e-3 * max(1.0, abs(popt))


def test_multi_kernel_and_fused_paths_agree(monkeypatch):
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(400, 1200, 8, seed=13)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SCS_AMD_FUSED", flag)
        res.append(capi.solve(amd, prob, verbose=0, acceleration_lookback=0, cg_tol_override=1e-12))
    assert res[0]["info"]["iter"] == res[1]["info"]["iter"]
    assert np.abs(res[0]["x"] - res[1]["x"]).max() <= 1e-7 * max(1.0, np.abs(res[0]["x"]).max())


def test_lp_only_and_zero_cone():
    ref = _ref("libscsindir_ref_exactcg.so")
    amd = capi.load("libscsamd.so")
    cone = dict(z=50, l=550)
    pr = problems.random_cone_prob(200, 600, 5, cone, seed=4)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, acceleration_lookback=0)
    ra, rr = capi.solve(amd, prob, cg_tol_override=1e-12, **kw), capi.solve(ref, prob, **kw)
    assert ra["info"]["iter"] == rr["info"]["iter"]
    assert _rel(ra["info"]["pobj"], rr["info"]["pobj"]) <= REL


def test_warm_start_and_update():
    ref = _ref("libscsindir_ref_exactcg.so")
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
        if name == "amd":
            lib.scs_amd_set_cg_tol_override(w, 1e-12)
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
    assert _rel(out["amd"][3], out["ref"][3]) <= REL
    assert out["amd"][1] < out["amd"][0]


def test_stepping_api_equals_scs_solve():
    amd = capi.load("libscsamd.so")
    T = amd._scs_types
    pr = problems.random_socp(300, 900, 8, seed=2)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    r1 = capi.solve(amd, prob, verbose=0, acceleration_lookback=0)
    st = capi.default_settings(amd, verbose=0, acceleration_lookback=0)
    w = amd.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    x, y, s = np.zeros(prob.n), np.zeros(prob.m), np.zeros(prob.m)
    sol = T.ScsSolution(x.ctypes.data_as(T.fp), y.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp))
    info = T.ScsInfo()
    assert amd.scs_amd_solve_begin(w, None, 0) == 0
    while not amd.scs_amd_solve_converged(w):
        assert amd.scs_amd_solve_steps(w, 7) >= 0
    amd.scs_amd_solve_end(w, C.byref(sol), C.byref(info))
    amd.scs_finish(w)
    assert info.iter == r1["info"]["iter"]
    assert np.array_equal(x, r1["x"])  # deterministic reductions: bit-identical reruns


@pytest.mark.parametrize("where", ["host", "dev"])
def test_anderson_acceleration_on_matches_reference(where, monkeypatch):
    """Default settings (AA type-I, lookback 10).  Host AA is pinned against src/aa.c in
    tests/test_aa_host.py, the device AA in tests/test_aa_dev_gpu.py; here the whole
    accelerated solve with either, exact CG on both sides."""
    monkeypatch.setenv("SCS_AMD_AA", where)
    ref = _ref("libscsindir_ref_exactcg.so")
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(600, 1800, 12, seed=21)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, eps_abs=1e-8, eps_rel=1e-8)  # long enough for the AA memory to fill (10 x 10 its)
    ra = capi.solve(amd, prob, cg_tol_override=1e-12, **kw)
    rr = capi.solve(ref, prob, **kw)
    ia, ir = ra["info"], rr["info"]
    assert ia["status_val"] == ir["status_val"] == 1
    assert ir["accepted_accel_steps"] > 0 and ia["accepted_accel_steps"] > 0
    assert abs(ia["accepted_accel_steps"] - ir["accepted_accel_steps"]) <= 3
    # AA's least-squares solve amplifies rounding differences: allow a different
    # termination check, but the optimum must agree
    assert abs(ia["iter"] - ir["iter"]) <= 50, (ia["iter"], ir["iter"])
    scale = max(1.0, abs(ir["pobj"]))
    assert abs(ia["pobj"] - ir["pobj"]) <= 5e-4 * scale
    # AA pays off on both sides vs. the un-accelerated run
    r0 = capi.solve(amd, prob, acceleration_lookback=0, cg_tol_override=1e-12, **kw)
    assert ia["iter"] <= r0["info"]["iter"]


def test_sdp_with_box_exact_cg_parity():
    ref = _ref("libscsindir_ref_exactcg.so")
    amd = capi.load("libscsamd.so")
    pr = problems.random_sdp(150, n_blocks=12, block=10, bsize=41, col_nnz=6, seed=5)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, acceleration_lookback=0)
    ra, rr = capi.solve(amd, prob, cg_tol_override=1e-12, **kw), capi.solve(ref, prob, **kw)
    ia, ir = ra["info"], rr["info"]
    assert ia["status_val"] == ir["status_val"] == 1
    assert ia["iter"] == ir["iter"]
    for k in ("pobj", "dobj", "res_pri", "res_dual", "gap"):
        assert _rel(ia[k], ir[k], floor=1e-3) <= REL, (k, ia[k], ir[k])


def test_sigint_during_solve_returns_scs_sigint():
    """src/ctrlc.c + src/scs.c:1400-1403: SIGINT while solving -> status SCS_SIGINT (-5), NaN-filled
    solution, and the previous handler is back afterwards."""
    import os
    import signal
    import threading

    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(20000, 40000, 10, seed=4)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    before = signal.getsignal(signal.SIGINT)
    t = threading.Timer(0.4, lambda: os.kill(os.getpid(), signal.SIGINT))
    t.start()
    try:
        r = capi.solve(amd, prob, verbose=0, eps_abs=1e-14, eps_rel=1e-14, max_iters=200000, acceleration_lookback=0)
    finally:
        t.cancel()
    assert r["info"]["status_val"] == -5 and r["info"]["status"] == "interrupted"
    assert r["info"]["iter"] == -1 and np.all(np.isnan(r["x"])) and np.all(np.isnan(r["s"]))
    assert signal.getsignal(signal.SIGINT) == before
    # and the library is usable afterwards
    r2 = capi.solve(amd, prob, verbose=0, max_iters=50)
    assert r2["info"]["status_val"] in (1, 2)


def test_concurrent_solves_from_several_host_threads_with_cg_graphs():
    """Workspaces are independent (reference include/scs.h:271-324: one caller thread per workspace,
    any number of workspaces).  Several host threads run scs_init / scs_solve at the same time; the
    PCG loops replay captured HIP graphs (these systems are below the graph threshold), so one
    thread allocates and uploads while another is inside a stream capture.  Every result must equal
    the single-threaded one bit for bit."""
    from concurrent.futures import ThreadPoolExecutor
    amd = capi.load("libscsamd.so")
    probs = []
    for sd in range(6):
        pr = problems.random_socp(2000 + 300 * sd, 6000 + 900 * sd, 8, seed=40 + sd)
        probs.append(capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"]))
    kw = dict(verbose=0, acceleration_lookback=0, max_iters=300)
    serial = [capi.solve(amd, p, **kw) for p in probs]
    for _ in range(2):
        with ThreadPoolExecutor(max_workers=4) as ex:
            par = list(ex.map(lambda p: capi.solve(amd, p, **kw), probs))
        for a, b in zip(serial, par):
            assert a["info"]["status_val"] == b["info"]["status_val"]
            assert a["info"]["iter"] == b["info"]["iter"]
            assert np.array_equal(a["x"], b["x"])


@pytest.mark.parametrize("seed", [0, 1])
def test_all_cone_types_exact_cg_trajectory_equals_the_restatement(seed):
    """Every cone type in one program (box, SOC, PSD, complex PSD, exponential and power, primal and
    dual): with exact CG the HIP solve walks the same trajectory as the plain-C restatement, which
    tests/test_oracle.py pins to the reference's on this same family."""
    import scipy.sparse as sp
    from oracle import pyoracle
    amd = capi.load("libscsamd.so")
    rng = np.random.default_rng(300 + seed)
    cone = dict(z=3, l=6, bu=[1.0, 2.0], bl=[-1.0, -0.5], q=[4, 7], s=[3, 5], cs=[3], ep=2, ed=2, p=[0.4, -0.7])
    m = capi.cone_rows(cone)
    n = m // 3
    z = rng.standard_normal(m)
    y = pyoracle.oracle_proj_dual_cone(cone, z)
    s = y - z
    x = rng.standard_normal(n)
    A = sp.random(m, n, density=min(1.0, 5.0 / n), random_state=seed, format="csc", data_rvs=rng.standard_normal)
    A = (A + sp.csc_matrix((np.full(n, 0.7), (np.arange(n), np.arange(n))), shape=(m, n))).tocsc()
    prob = capi.Problem(A, A @ x + s, -(A.T @ y), cone)
    kw = dict(eps_abs=1e-5, eps_rel=1e-5, max_iters=30000)
    ra = capi.solve(amd, prob, verbose=0, acceleration_lookback=0, cg_tol_override=1e-12, **kw)
    ro = pyoracle.oracle_solve(prob, cg_tol_override=1e-12, **kw)
    assert ra["info"]["status_val"] == ro["info"]["status_val"] == 1
    assert ra["info"]["iter"] == ro["info"]["iter"]
    assert ra["info"]["scale_updates"] == ro["info"]["scale_updates"]
    for key in ("pobj", "dobj", "res_pri", "res_dual", "gap", "scale"):
        assert abs(ra["info"][key] - ro["info"][key]) <= 1e-6 * max(abs(ro["info"][key]), 1e-3), key
    assert np.abs(ra["x"] - ro["x"]).max() <= 1e-6 * max(1.0, np.abs(ro["x"]).max())


def test_verbose_output_has_the_reference_layout(capfd):
    """`verbose=1` prints the reference's tables (src/scs.c:113-272): same settings block, same column header,
    same row and footer formats -- only the banner and the lin-sys name identify this backend."""
    import re
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built")
    ref = pyoracle.load_ref()
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(60, m=150, col_nnz=5, seed=3)
    outs = {}
    for name, lib in (("ref", ref), ("amd", amd)):
        prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"], T=lib._scs_types)
        capfd.readouterr()
        capi.solve(lib, prob, verbose=1, max_iters=100, acceleration_lookback=0)
        import ctypes as C
        C.CDLL(None).fflush(None)
        outs[name] = capfd.readouterr().out.splitlines()
    r, a = outs["ref"], outs["amd"]
    rule = "-" * 66
    assert r.count(rule) == a.count(rule) == 7
    for key in ("problem:  variables n:", "settings: eps_abs:", "\t  alpha:", "\t  max_iters:", " iter | pri res"):
        lr = [l for l in r if l.startswith(key)]
        la = [l for l in a if l.startswith(key)]
        assert lr and lr == la, key
    assert [l for l in a if l.startswith("lin-sys:  ")] and [l for l in a if l.startswith("\t  nnz(A): ")] == \
        [l for l in r if l.startswith("\t  nnz(A): ")]
    row = re.compile(r"^ *\d+\|( *-?\d\.\d\de[+-]\d\d ){6}$")
    rows_r, rows_a = [l for l in r if row.match(l)], [l for l in a if row.match(l)]
    assert len(rows_a) == len(rows_r) >= 2
    assert [l.split("|")[0] for l in rows_a] == [l.split("|")[0] for l in rows_r]
    for key, pat in (("status:  ", None), ("timings: total: ", r"^timings: total: \d\.\d\de[+-]\d\ds = setup: \d\.\d\de[+-]\d\ds \+ solve: \d\.\d\de[+-]\d\ds$"),
                     ("\t lin-sys: ", r"^\t lin-sys: \d\.\d\de[+-]\d\ds, cones: \d\.\d\de[+-]\d\ds, accel: \d\.\d\de[+-]\d\ds$"),
                     ("objective = ", r"^objective = -?\d+\.\d{6}( \(inaccurate\))?$")):
        la = [l for l in a if l.startswith(key)]
        assert len(la) == 1 and len([l for l in r if l.startswith(key)]) == 1, key
        if pat:
            assert re.match(pat, la[0]), la[0]


@pytest.mark.parametrize("shape", [(1000, 3000, 32), (400, 1200, 8), (1023, 2100, 10), (777, 1500, 6)])
def test_two_launch_cg_path_is_bit_identical_to_the_four_kernel_path(shape, monkeypatch):
    """n <= 1024: a CG iteration is k_cg2_a (update + direction of the previous iteration, redundantly in every
    workgroup, then the A product out of LDS) + the transposed product instead of four kernels (linsys.hip).  Same
    arithmetic down to the shape of every partial sum, so the whole solve -- default inexact CG schedule, where
    any rounding difference would change iteration counts -- is bit-identical; also with graph replay off."""
    amd = capi.load("libscsamd.so")
    n, m, cn = shape
    pr = problems.random_socp(n, m, cn, seed=n)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, acceleration_lookback=0, max_iters=150, eps_abs=1e-9, eps_rel=1e-9, want_stats=True)
    monkeypatch.setenv("SCS_AMD_FUSED", "0")
    outs = {}
    for name, env in (("four", dict(SCS_AMD_CG2="0")), ("two", dict(SCS_AMD_CG2="1")),
                      ("two_nograph", dict(SCS_AMD_CG2="1", SCS_AMD_GRAPH="0"))):
        for k in ("SCS_AMD_CG2", "SCS_AMD_GRAPH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[name] = capi.solve(amd, prob, **kw)
    for a, b in (("four", "two"), ("four", "two_nograph")):
        ra, rb = outs[a], outs[b]
        assert ra["stats"]["cg_iters"] == rb["stats"]["cg_iters"] > 0, (a, b)
        assert ra["info"]["iter"] == rb["info"]["iter"]
        for v in ("x", "y", "s"):
            assert np.array_equal(ra[v], rb[v]), (a, b, v, np.abs(ra[v] - rb[v]).max())
