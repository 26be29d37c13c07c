"""B1 parity: our HIP scs_solve_lin_sys vs the reference's CPU indirect backend
(linsys/cpu/indirect/private.c:284-324) through the SAME C ABI on the same inputs."""
import ctypes as C

import numpy as np
import pytest

from scs_amd import capi
from tests import probgen

pytestmark = pytest.mark.gpu


def _ref():
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return pyoracle.load_ref()


def _solve_with(lib, matA, matP, dr, b, s, tol):
    T = lib._scs_types
    w = lib.scs_init_lin_sys_work(C.byref(matA), C.byref(matP) if matP is not None else None,
                                  dr.ctypes.data_as(T.fp))
    assert w
    out = b.copy()
    rc = lib.scs_solve_lin_sys(w, out.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp) if s is not None else None, tol)
    assert rc == 0
    return w, out


@pytest.mark.parametrize("n,m,col_nnz,warm,tol", [
    (50, 150, 4, False, 1e-12),
    (1000, 3000, 32, False, 1e-12),
    (1000, 3000, 32, True, 1e-7),
    (3000, 7001, 9, True, 1e-4),
    (20000, 50000, 10, False, 1e-9),
])
def test_solve_lin_sys_matches_reference(n, m, col_nnz, warm, tol):
    ref = _ref()
    amd = capi.load("libscsamd_linsys.so")
    rng = np.random.default_rng(n + m)
    A = probgen.random_csc(m, n, col_nnz, seed=7)
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
    dr = probgen.diag_r(n, m, z=m // 10)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n) * 0.1 if warm else None
    wr, xr = _solve_with(ref, prob.matA, None, dr, b, s, tol)
    wa, xa = _solve_with(amd, prob.matA, None, dr, b, s, tol)
    # both solved the same SPD system to `tol` in the residual inf-norm; at
    # tol=1e-12 they agree to rounding, otherwise to a small multiple of tol / lambda_min
    scale = np.abs(xr).max()
    if tol <= 1e-9:
        assert np.abs(xa - xr).max() <= 1e-7 * scale
    # exact check independent of CG path: KKT residual of our answer
    x, y = xa[:n], xa[n:]
    Asp = prob.sparse()
    r1 = dr[:n] * x + Asp.T @ y - b[:n]
    r2 = Asp @ x - dr[n:] * y - b[n:]
    # reduced residual G x - rhs
    red = r1 + Asp.T @ (r2 / dr[n:])
    assert np.abs(red).max() < max(tol, 1e-12) * 1.01 + 1e-10 * np.abs(b).max()
    assert np.abs(r2).max() < 1e-9 * max(1.0, np.abs(b).max()) * dr[n:].max()
    # second solve on the same workspace + diag_r update
    dr2 = probgen.diag_r(n, m, z=m // 10, scale=2.5)
    assert amd.scs_update_lin_sys_diag_r(wa, dr2.ctypes.data_as(capi.T64.fp)) == 0
    assert ref.scs_update_lin_sys_diag_r(wr, dr2.ctypes.data_as(capi.T64.fp)) == 0
    b2 = rng.uniform(-1, 1, n + m)
    o1, o2 = b2.copy(), b2.copy()
    assert amd.scs_solve_lin_sys(wa, o1.ctypes.data_as(capi.T64.fp), None, 1e-12) == 0
    assert ref.scs_solve_lin_sys(wr, o2.ctypes.data_as(capi.T64.fp), None, 1e-12) == 0
    assert np.abs(o1 - o2).max() <= 1e-7 * np.abs(o2).max()
    amd.scs_free_lin_sys_work(wa)
    ref.scs_free_lin_sys_work(wr)


def test_zero_rhs_short_circuit():
    amd = capi.load("libscsamd_linsys.so")
    n, m = 40, 100
    A = probgen.random_csc(m, n, 3, seed=1)
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
    dr = probgen.diag_r(n, m, z=10)
    b = np.full(n + m, 1e-13)
    w, out = _solve_with(amd, prob.matA, None, dr, b, None, 1e-9)
    assert np.all(out == 0.0)  # private.c:296-299
    amd.scs_free_lin_sys_work(w)


def test_with_P_matches_reference():
    ref = _ref()
    amd = capi.load("libscsamd_linsys.so")
    import scipy.sparse as sp
    n, m = 300, 500
    rng = np.random.default_rng(3)
    A = probgen.random_csc(m, n, 5, seed=11)
    B = sp.random(n, n, density=0.02, random_state=5, format="csc")
    P = (B @ B.T + sp.identity(n) * 0.1).tocsc()
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m), P=P)
    dr = probgen.diag_r(n, m, z=50)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n)
    wr, xr = _solve_with(ref, prob.matA, prob.matP, dr, b, s, 1e-12)
    wa, xa = _solve_with(amd, prob.matA, prob.matP, dr, b, s, 1e-12)
    assert np.abs(xa - xr).max() <= 1e-8 * np.abs(xr).max()
    amd.scs_free_lin_sys_work(wa)
    ref.scs_free_lin_sys_work(wr)


def test_wave_rows_spmv_path_equals_reference_and_csr_stream_path(monkeypatch):
    """The wave-owned-rows kernel (spmv_wave.h, used when the gathered vector overflows L2) forced on
    at a size the reference backend solves in a second: same answer as the reference's own
    linsys/cpu/indirect and as the CSR-stream kernel; small nonzero budgets force several units per
    wave, single-row units and the grid-stride loop."""
    amd = capi.load("libscsamd_linsys.so")
    from oracle import pyoracle
    ref = pyoracle.load_ref() if pyoracle.ref_available() else None
    n, m = 30000, 70001
    rng = np.random.default_rng(9)
    A = probgen.random_csc(m, n, 7, seed=5)
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
    dr = probgen.diag_r(n, m, z=m // 10)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n)
    outs = []
    for flag, budget in (("0", None), ("1", None), ("1", "64"), ("1", "100000")):
        monkeypatch.setenv("SCS_AMD_WAVEROWS", flag)
        if budget:
            monkeypatch.setenv("SCS_AMD_WR_NNZ", budget)
        else:
            monkeypatch.delenv("SCS_AMD_WR_NNZ", raising=False)
        w, out = _solve_with(amd, prob.matA, None, dr, b, s, 1e-12)
        amd.scs_free_lin_sys_work(w)
        outs.append(out)
    for o in outs[1:]:
        assert np.abs(outs[0] - o).max() <= 1e-9 * np.abs(outs[0]).max()
    if ref is not None:
        wr, xr = _solve_with(ref, prob.matA, None, dr, b, s, 1e-12)
        ref.scs_free_lin_sys_work(wr)
        for o in outs:
            assert np.abs(o - xr).max() <= 1e-8 * np.abs(xr).max()


def test_small_n_very_tall_matrix_with_wave_rows_layout(monkeypatch):
    """n <= 1024 used to take the two-launch CG path (k_cg2_a) even when the wave-owned-rows layout
    was built (nnz >= 1e6, or forced), and then summed the wrong number of p'Gp partials.  A very tall A with few
    columns: forced wave rows + forced two-launch request must give the reference's answer and the CSR-stream path's."""
    amd = capi.load("libscsamd_linsys.so")
    from oracle import pyoracle
    ref = pyoracle.load_ref() if pyoracle.ref_available() else None
    n, m = 1000, 100000
    rng = np.random.default_rng(21)
    A = probgen.random_csc(m, n, 120, seed=8)
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
    dr = probgen.diag_r(n, m, z=m // 10)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n)
    outs = {}
    for wave, cg2 in (("0", "0"), ("0", "1"), ("1", "1"), ("1", "0")):
        monkeypatch.setenv("SCS_AMD_WAVEROWS", wave)
        monkeypatch.setenv("SCS_AMD_CG2", cg2)
        monkeypatch.setenv("SCS_AMD_WR_NNZ", "3000")
        w, out = _solve_with(amd, prob.matA, None, dr, b, s, 1e-12)
        amd.scs_free_lin_sys_work(w)
        outs[(wave, cg2)] = out
    base = outs[("0", "0")]
    assert np.array_equal(base, outs[("0", "1")])  # two-launch path is bit-identical to the four-kernel path
    for k, o in outs.items():
        assert np.abs(o - base).max() <= 1e-9 * np.abs(base).max(), k
    if ref is not None:
        wr, xr = _solve_with(ref, prob.matA, None, dr, b, s, 1e-12)
        ref.scs_free_lin_sys_work(wr)
        for k, o in outs.items():
            assert np.abs(o - xr).max() <= 1e-8 * np.abs(xr).max(), k


def test_banded_matrix_takes_the_pipelined_wave_kernel_and_matches_reference(monkeypatch, capfd):
    """A matrix with column locality (scs_amd/problems.py banded_rows): WaveRowsDev::build measures < 0.5 distinct
    lines of the gathered vector per entry and picks the software-pipelined instantiation of csr_wave_kernel; same
    answers as the reference backend and as the plain instantiation (forced)."""
    from scs_amd import problems
    amd = capi.load("libscsamd_linsys.so")
    from oracle import pyoracle
    ref = pyoracle.load_ref() if pyoracle.ref_available() else None
    n, m = 40000, 80000
    A = problems.random_cone_prob(n, m, 8, dict(l=m), seed=13, band=1024)["A"]
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
    dr = probgen.diag_r(n, m, z=m // 10)
    rng = np.random.default_rng(4)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n)
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")
    monkeypatch.setenv("SCS_AMD_DEBUG", "1")
    outs = []
    for pipe in (None, "0", "1", "2"):   # 2 (round 6): two chunks per round trip, the mid-size instantiation
        if pipe is None:
            monkeypatch.delenv("SCS_AMD_WR_PIPE", raising=False)
        else:
            monkeypatch.setenv("SCS_AMD_WR_PIPE", pipe)
        w, out = _solve_with(amd, prob.matA, None, dr, b, s, 1e-12)
        amd.scs_free_lin_sys_work(w)
        outs.append(out)
        if pipe is None:
            err = capfd.readouterr().err
            assert err.count("-> pipelined stream") == 2, err[-800:]   # both orientations detected the locality
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)                              # same entry order: every instantiation agrees bit for bit
    if ref is not None:
        wr, xr = _solve_with(ref, prob.matA, None, dr, b, s, 1e-12)
        ref.scs_free_lin_sys_work(wr)
        assert np.abs(outs[0] - xr).max() <= 1e-8 * np.abs(xr).max()
    # units of ODD chunk counts (a last trip with one chunk) and of a single short chunk: two-chunk trips against the plain kernel
    for unit_nnz in ("700", "130"):
        monkeypatch.setenv("SCS_AMD_WR_NNZ", unit_nnz)
        pair = []
        for pipe in ("0", "2"):
            monkeypatch.setenv("SCS_AMD_WR_PIPE", pipe)
            w, out = _solve_with(amd, prob.matA, None, dr, b, s, 1e-12)
            amd.scs_free_lin_sys_work(w)
            pair.append(out)
        assert np.array_equal(pair[0], pair[1]), unit_nnz


@pytest.mark.parametrize("wpb,bars,unit_nnz,wpc", [("16", "4", "600", None), ("16", "1", "1500", "1"), ("8", "4", "300", "2")])
def test_lockstep_wave_kernel_matches_the_plain_one_and_the_reference(wpb, bars, unit_nnz, wpc, monkeypatch):
    """Round 4: csr_wave_lockstep_kernel (one workgroup of 16 or 8 waves per CU issuing its gather instructions together, chunks
    stored by quarter-windows; the library's choice for fp64 systems from 5e6 nonzeros on).  Forced on here at a size where the
    reference finishes in seconds, with a unit budget small enough that the grid takes SEVERAL rounds and the waves of a workgroup
    get units of different chunk counts (uneven rows: the matrix is tall with a few dense rows appended) -- a wave that runs out keeps
    the others company at the barriers.  Same answer as the plain kernel (to rounding: the chunk order differs), as the csr-stream
    kernel, and as the reference backend; identical bits from run to run."""
    import scipy.sparse as sp
    amd = capi.load("libscsamd_linsys.so")
    n, m = 30000, 70001
    A = probgen.random_csc(m - 40, n, 9, seed=21)
    dense = sp.random(40, n, density=0.05, random_state=3, format="csc")   # 40 rows of ~1500 entries each: units of very different length
    A = sp.vstack([A, dense]).tocsc()
    A.sort_indices()
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
    dr = probgen.diag_r(n, m, z=m // 10)
    rng = np.random.default_rng(4)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n) * 0.1
    outs = {}
    for name, env in (("stream", dict(SCS_AMD_WAVEROWS="0")),
                      ("plain", dict(SCS_AMD_WAVEROWS="1", SCS_AMD_WR_LOCKSTEP="0", SCS_AMD_WR_NNZ=unit_nnz)),
                      ("lockstep", dict(SCS_AMD_WAVEROWS="1", SCS_AMD_WR_LOCKSTEP="1", SCS_AMD_WR_LS_WPB=wpb, SCS_AMD_WR_LS_BARRIERS=bars,
                                        SCS_AMD_WR_NNZ=unit_nnz)),
                      ("lockstep_again", dict(SCS_AMD_WAVEROWS="1", SCS_AMD_WR_LOCKSTEP="1", SCS_AMD_WR_LS_WPB=wpb, SCS_AMD_WR_LS_BARRIERS=bars,
                                              SCS_AMD_WR_NNZ=unit_nnz))):
        for k in ("SCS_AMD_WAVEROWS", "SCS_AMD_WR_LOCKSTEP", "SCS_AMD_WR_LS_WPB", "SCS_AMD_WR_LS_BARRIERS", "SCS_AMD_WR_NNZ"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if wpc and name.startswith("lockstep"):
            monkeypatch.setenv("SCS_AMD_WR_WPC", wpc)   # few resident waves per CU: the grid walks its units in several rounds
        else:
            monkeypatch.delenv("SCS_AMD_WR_WPC", raising=False)
        w, out = _solve_with(amd, prob.matA, None, dr, b, s, 1e-12)
        amd.scs_free_lin_sys_work(w)
        outs[name] = out
    scale = np.abs(outs["stream"]).max()
    assert np.array_equal(outs["lockstep"], outs["lockstep_again"])            # deterministic
    assert np.abs(outs["lockstep"] - outs["plain"]).max() <= 1e-9 * scale
    assert np.abs(outs["lockstep"] - outs["stream"]).max() <= 1e-9 * scale
    ref = _ref()
    wr, xr = _solve_with(ref, prob.matA, None, dr, b, s, 1e-12)
    ref.scs_free_lin_sys_work(wr)
    assert np.abs(outs["lockstep"] - xr).max() <= 1e-8 * np.abs(xr).max()


def _stats(lib, w):
    st = lib._scs_types.ScsAmdStats()
    lib.scs_amd_linsys_get_stats(w, C.byref(st))
    return {k: getattr(st, k) for k, _ in lib._scs_types.ScsAmdStats._fields_}


@pytest.mark.parametrize("with_p", [False, True])
def test_three_kernel_cg_iteration_matches_the_four_kernel_one_and_the_reference(monkeypatch, with_p):
    """Round 5: k_cg3_update does the stop test, alpha, beta and the four vector updates of private.c:181-214 in ONE launch; beta comes
    from z'r - 2 alpha z'Gp + alpha^2 Gp'MGp (three dot products of the transposed product's epilogue) instead of a second reduction.
    Same answers as the four-kernel iteration and as the reference's linsys/cpu/indirect at a tight tolerance; at loose tolerances the
    iteration counts within a few percent and a residual below the tolerance (the returned iterate is the first with |r|_inf < tol)."""
    import scipy.sparse as sp
    amd = capi.load("libscsamd_linsys.so")
    from oracle import pyoracle
    ref = pyoracle.load_ref() if pyoracle.ref_available() else None
    n, m = 30000, 70001
    rng = np.random.default_rng(19)
    A = probgen.random_csc(m, n, 7, seed=15)
    P = None
    if with_p:
        Pm = sp.random(n, n, density=2.0 / n, random_state=3, format="csc")
        P = sp.triu(Pm @ Pm.T + sp.identity(n) * 0.5, format="csc")
        P.sort_indices()
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m), P=P)
    dr = probgen.diag_r(n, m, z=m // 10)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n)
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")
    monkeypatch.setenv("SCS_AMD_GRAPH", "0")
    res = {}
    for tol in (1e-12, 1e-6, 1e-2):
        for cg3 in ("0", "1"):
            monkeypatch.setenv("SCS_AMD_CG3", cg3)
            w, out = _solve_with(amd, prob.matA, prob.matP if with_p else None, dr, b, s, tol)
            res[tol, cg3] = (out, _stats(amd, w)["cg_iters"])
            amd.scs_free_lin_sys_work(w)
        (x4, it4), (x3, it3) = res[tol, "0"], res[tol, "1"]
        # (the expanded beta moves every search direction a little: the two loops reach a tolerance within a few percent of the same
        # iteration, not at the same one -- measured 2298 vs 2310 at 1e-12, 1276 vs 1314 at 1e-6; one of the reasons it is opt-in)
        assert it3 > 0 and abs(it3 - it4) <= 0.06 * it4, (tol, it3, it4)
        scale = np.abs(x4).max()
        assert np.abs(x3 - x4).max() <= (1e-9 if tol == 1e-12 else 10 * tol) * scale, (tol, np.abs(x3 - x4).max())
        # the stopping rule itself: |b_x + A' R_y^-1 b_y - (R_x + P + A' R_y^-1 A) x|_inf < tol for the returned x
        # (recomputed on the host; the recurred residual the solver tests differs from it by rounding: skipped at the 1e-12 floor)
        if tol > 1e-12:
            As = prob.sparse()
            Ry = dr[n:]
            Pf = (P + P.T - sp.diags(P.diagonal())) if with_p else None
            G = lambda v: dr[:n] * v + As.T @ ((As @ v) / Ry) + ((Pf @ v) if with_p else 0)
            rhs = b[:n] + As.T @ (b[n:] / Ry)
            assert np.abs(rhs - G(x3[:n])).max() < tol * 1.001
    if ref is not None:
        wr, xr = _solve_with(ref, prob.matA, prob.matP if with_p else None, dr, b, s, 1e-12)
        ref.scs_free_lin_sys_work(wr)
        assert np.abs(res[1e-12, "1"][0] - xr).max() <= 1e-8 * np.abs(xr).max()


@pytest.mark.parametrize("libname", ["libscsamd_linsys.so", "libscsamd_f32.so"])
def test_wide_wave_layout_equals_the_narrow_one_bit_for_bit(libname, monkeypatch):
    """Round 6: beyond 2^26 rows or columns the 32-bit packed word has no room for the local row; the WIDE layout keeps the column in
    the word and the local row in a 16-bit array of its own (csr_wave_wide_kernel, spmv_wave.h) -- found by the nnz = 2.2e9 run, which
    had fallen back to the CSR-stream kernel.  Forced here at a small size (option wr_wide): both builders agree byte for byte (verify
    mode), and a linear solve through it returns the SAME BITS as csr_wave_kernel<EPI, 0> on the narrow layout (same entry order, same
    adds) and the reference's answer; ragged units and a few long rows included."""
    import scipy.sparse as sp
    amd = capi.load(libname)
    T = amd._scs_types
    n, m = 30000, 70001
    A = probgen.random_csc(m - 40, n, 9, seed=23)
    dense = sp.random(40, n, density=0.05, random_state=5, format="csc")
    A = sp.vstack([A, dense]).tocsc()
    A.sort_indices()
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m), T=T)
    f = T.np_float
    dr = probgen.diag_r(n, m, z=m // 10).astype(f)
    rng = np.random.default_rng(6)
    b = rng.uniform(-1, 1, n + m).astype(f)
    s = (rng.uniform(-1, 1, n) * 0.1).astype(f)
    tol = 1e-12 if f is np.float64 else 1e-5
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")
    monkeypatch.setenv("SCS_AMD_WR_LOCKSTEP", "0")
    monkeypatch.setenv("SCS_AMD_WR_PIPE", "0")
    monkeypatch.setenv("SCS_AMD_WR_BUILD", "verify")
    outs = {}
    for wide, unit_nnz in (("0", "1500"), ("1", "1500"), ("0", "300"), ("1", "300")):
        monkeypatch.setenv("SCS_AMD_WR_WIDE", wide)
        monkeypatch.setenv("SCS_AMD_WR_NNZ", unit_nnz)
        w = amd.scs_init_lin_sys_work(C.byref(prob.matA), None, dr.ctypes.data_as(T.fp))
        assert w, (wide, unit_nnz)
        out = b.copy()
        assert amd.scs_solve_lin_sys(w, out.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp), tol) == 0
        amd.scs_free_lin_sys_work(w)
        outs[(wide, unit_nnz)] = out
    assert np.array_equal(outs[("0", "1500")], outs[("1", "1500")])
    assert np.array_equal(outs[("0", "300")], outs[("1", "300")])
    assert np.abs(outs[("1", "1500")]).max() > 0
    if f is np.float64:
        ref = _ref()
        wr, xr = _solve_with(ref, prob.matA, None, dr, b, s, 1e-12)
        ref.scs_free_lin_sys_work(wr)
        assert np.abs(outs[("1", "1500")] - xr).max() <= 1e-8 * np.abs(xr).max()
