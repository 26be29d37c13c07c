"""B1' drop-in: `_scs_proj_dual_cone` of libscsamd_cones.so against the reference's own
`_scs_proj_dual_cone` (golden vectors generated from the reference build,
tests/golden/cones.npz) through the reference's calling convention: lazily normalised box
bounds via `scal->D`, `r_y` metric, NULL/NULL as the reference's test helpers call it."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from scs_amd import capi

pytestmark = pytest.mark.gpu
T = capi.T64
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class ScsScaling(C.Structure):  # include/scs_amd.h (reference include/scs_work.h:24-29)
    _fields_ = [("D", T.fp), ("E", T.fp), ("m", C.c_int), ("n", C.c_int), ("primal_scale", C.c_double),
                ("dual_scale", C.c_double)]


def _lib():
    l = C.CDLL(capi.lib_path("libscsamd_cones.so"))
    l._scs_init_cone.restype = C.c_void_p
    l._scs_init_cone.argtypes = [C.POINTER(T.ScsCone), C.c_int]
    l._scs_finish_cone.argtypes = [C.c_void_p]
    l._scs_proj_dual_cone.argtypes = [T.fp, C.c_void_p, C.POINTER(ScsScaling), T.fp]
    return l


def test_proj_dual_cone_matches_reference_golden_vectors():
    lib = _lib()
    g = np.load(os.path.join(G, "cones.npz"))
    meta = json.load(open(os.path.join(G, "cones_meta.json")))
    assert len(meta) >= 8
    for name, cone in meta.items():
        k = capi.make_cone(cone)
        m = capi.cone_rows(cone)
        c = lib._scs_init_cone(C.byref(k), m)
        assert c, name
        tol = 1e-11 if ("psd" in name or name in ("mixed", "all", "all_c")) else 1e-12
        for rep in range(2):  # the second round reuses the device workspace (box warm start included)
            for variant in ("eucl", "ry"):
                x = np.array(g[f"{name}_{variant}_x"])
                want = g[f"{name}_{variant}_y"]
                r = np.array(g[f"{name}_{variant}_r"]) if variant == "ry" else None
                rc = lib._scs_proj_dual_cone(x.ctypes.data_as(T.fp), c, None,
                                             r.ctypes.data_as(T.fp) if r is not None else None)
                assert rc == 0, name
                err = np.abs(x - want).max() / max(1.0, np.abs(want).max())
                assert err <= tol, (name, variant, rep, err)
        lib._scs_finish_cone(c)


def test_box_bounds_are_normalised_lazily_from_scal():
    """src/cones.c:1557-1565 + 1161-1177: with `scal` the box bounds are rescaled by D[j+1]/D[0]
    at the first projection -- equal to projecting with pre-scaled bounds and no `scal`."""
    lib = _lib()
    rng = np.random.default_rng(0)
    nb = 40
    bu, bl = rng.uniform(0.5, 2.0, nb), -rng.uniform(0.5, 2.0, nb)
    D = rng.uniform(0.2, 3.0, nb + 1)
    x0 = rng.standard_normal(nb + 1) * 3
    outs = []
    for cone, scal in ((dict(bu=bu, bl=bl), ScsScaling(D.ctypes.data_as(T.fp), None, nb + 1, 0, 1.0, 1.0)),
                       (dict(bu=bu * D[1:] / D[0], bl=bl * D[1:] / D[0]), None)):
        k = capi.make_cone(cone)
        c = lib._scs_init_cone(C.byref(k), nb + 1)
        x = x0.copy()
        assert lib._scs_proj_dual_cone(x.ctypes.data_as(T.fp), c, C.byref(scal) if scal else None, None) == 0
        outs.append(x)
        lib._scs_finish_cone(c)
    assert np.array_equal(np.asarray(k.bu[0:1]), np.asarray(k.bu[0:1]))  # caller's cone untouched (no mutation)
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-13, atol=1e-13)
    assert np.abs(outs[0] - x0).max() > 1e-3


def test_psd_warm_start_stays_accurate_over_a_drifting_sequence():
    """The PSD kernel carries each block's eigenbasis between calls.  Over a
    slowly drifting sequence -- 150 calls, i.e. across two cold restarts -- every projection must
    still match an independent numpy eigendecomposition; a jump to an unrelated matrix must too."""
    from scs_amd import problems
    lib = capi.load("libscsamd.so")
    cone = dict(l=3, s=[50, 7, 1, 24])
    m = capi.cone_rows(cone)
    k = capi.make_cone(cone)
    w = lib.scs_amd_cone_init(C.byref(k), m, None)
    assert w
    rng = np.random.default_rng(8)
    base, vel = rng.standard_normal(m), rng.standard_normal(m)
    worst = 0.0
    for it in range(150):
        x = base + 0.01 * it * vel + 1e-3 * rng.standard_normal(m)
        if it == 100:
            x = rng.standard_normal(m) * 5  # unrelated input: the carried basis is useless, not harmful
        want = problems.proj_dual_cone_np(x, cone)
        got = x.copy()
        assert lib.scs_amd_cone_proj_dual(w, got.ctypes.data_as(T.fp), None) == 0
        worst = max(worst, np.abs(got - want).max() / max(1.0, np.abs(want).max()))
    lib.scs_amd_cone_finish(w)
    assert worst <= 1e-11, worst


def _ref_lib():
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built")
    return pyoracle.load_ref()


@pytest.mark.parametrize("nb", [1000000, 20000])
def test_large_box_cone_multi_workgroup_path_matches_reference(nb, monkeypatch):
    """A box of 1e6 rows (every variable-bounded LP/QP from CVXPY has one) runs its Newton iteration as
    chip-wide launches (k_box_step / k_box_apply, cones.hip) instead of one workgroup: same projection as
    the reference's src/cones.c:1182-1245 with infinite bounds and the R_y metric, repeated calls
    (warm-started t), and equal to the one-workgroup kernel."""
    ref = _ref_lib()
    lib = _lib()
    rng = np.random.default_rng(5)
    bu, bl = rng.uniform(0.1, 2.0, nb), -rng.uniform(0.1, 2.0, nb)
    bu[::7] = 1e20   # |.| >= 1e15 means infinite (cones.c:1167-1175)
    bl[::11] = -1e20
    r_y = rng.uniform(0.5, 3.0, nb + 1)
    # the reference turns |bound| >= 1e15 into +-inf only inside normalize_box_cone, i.e. only when a scaling is
    # passed (cones.c:1561, which also rescales the caller's arrays in place: every library gets its own copy);
    # without one 1e20 stays a finite bound.  Both behaviours are compared.
    D = rng.uniform(0.5, 2.0, nb + 1)
    Tr = ref._scs_types
    ref._scs_proj_dual_cone.argtypes = [Tr.fp, C.c_void_p, C.c_void_p, Tr.fp]
    outs = {}
    for use_scal in (True, False):
        kr = capi.make_cone(dict(bu=bu.copy(), bl=bl.copy()), Tr)
        wr = ref._scs_init_cone(C.byref(kr), nb + 1)
        scal = ScsScaling(D.ctypes.data_as(T.fp), None, nb + 1, 0, 1.0, 1.0) if use_scal else None
        for mode in ("1", "0"):
            monkeypatch.setenv("SCS_AMD_BOX_MULTI", mode)
            k = capi.make_cone(dict(bu=bu.copy(), bl=bl.copy()))
            c = lib._scs_init_cone(C.byref(k), nb + 1)
            assert c
            res = []
            for rep in range(3):
                x0 = np.random.default_rng(100 + rep).standard_normal(nb + 1) * 2.0
                x0[0] = abs(x0[0]) * (0.2 if rep != 1 else -1.0)  # rep 1: t < 0 exercises the clamp at zero
                for r in (None, r_y):
                    x = x0.copy()
                    assert lib._scs_proj_dual_cone(x.ctypes.data_as(T.fp), c, C.byref(scal) if scal else None,
                                                   r.ctypes.data_as(T.fp) if r is not None else None) == 0
                    res.append(x)
                    if mode == "1":
                        want = x0.copy()
                        assert ref._scs_proj_dual_cone(want.ctypes.data_as(Tr.fp), wr, C.byref(scal) if scal else None,
                                                       r.ctypes.data_as(Tr.fp) if r is not None else None) == 0
                        err = np.abs(x - want).max() / max(1.0, np.abs(want).max())
                        assert err <= 1e-11, (nb, use_scal, rep, r is not None, err)
            outs[(use_scal, mode)] = res
            lib._scs_finish_cone(c)
        ref._scs_finish_cone(wr)
    for use_scal in (True, False):
        for a, b in zip(outs[(use_scal, "1")], outs[(use_scal, "0")]):
            assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("cone", [dict(s=[100, 10, 131]), dict(s=[200], cs=[60]), dict(s=[300])])
def test_psd_blocks_beyond_the_lds_path_match_reference(cone):
    """Blocks of order > 92 (A and V no longer fit one CU's LDS) run the Jacobi steps as chip-wide launches
    (psd_big.h), next to small blocks that stay in LDS; complex Hermitian blocks through their real embedding
    (order 2 * 60 = 120).  Same projection as the reference's LAPACK path (src/cones.c:999-1155) to 1e-11, twice
    (the second call reuses the device workspace)."""
    ref = _ref_lib()
    lib = _lib()
    Tr = ref._scs_types
    m = capi.cone_rows(cone)
    kr = capi.make_cone(cone, Tr)
    wr = ref._scs_init_cone(C.byref(kr), m)
    k = capi.make_cone(cone)
    c = lib._scs_init_cone(C.byref(k), m)
    assert c and wr
    for rep in range(2):
        x0 = np.random.default_rng(7 + rep).standard_normal(m)
        got, want = x0.copy(), x0.copy()
        assert lib._scs_proj_dual_cone(got.ctypes.data_as(T.fp), c, None, None) == 0
        assert ref._scs_proj_dual_cone(want.ctypes.data_as(Tr.fp), wr, None, None) == 0
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert err <= 1e-11, (cone, rep, err)
        assert np.abs(got - x0).max() > 1e-3
    lib._scs_finish_cone(c)
    ref._scs_finish_cone(wr)


def _pack_psd(X):
    """full symmetric k x k -> SCS's packed lower triangle, column-major, off-diagonals * sqrt(2) (src/cones.c:1018-1025)"""
    k = X.shape[0]
    out = []
    for j in range(k):
        col = X[j:, j].copy()
        col[1:] *= np.sqrt(2.0)
        out.append(col)
    return np.concatenate(out)


def _unpack_psd(v, k):
    X = np.zeros((k, k))
    o = 0
    for j in range(k):
        col = v[o:o + k - j].copy()
        col[1:] /= np.sqrt(2.0)
        X[j:, j] = col
        X[j, j:] = col
        o += k - j
    return X


@pytest.mark.parametrize("k", [512, 640])
def test_large_psd_block_with_a_wide_eigenvalue_spread(k):
    """the blocked Jacobi update used to compute tile (P, Q) and tile (Q, P) with independent matrix-core products
    (symmetric only to rounding) while the inner sweep reads one triangle; round 4 writes the mirror tile from the same product.
    A block of order >= 512 whose eigenvalues span 12 decades on both signs: the projection (dual cone of the PSD cone = the
    PSD cone, r_y = NULL) must equal V max(L, 0) V' to 1e-9 of |A|, be exactly what a second call returns, and a projected
    matrix must be a fixed point."""
    lib = _lib()
    rng = np.random.default_rng(k)
    Q, _ = np.linalg.qr(rng.standard_normal((k, k)))
    lam = np.concatenate([10.0 ** rng.uniform(-8, 4, k // 2), -(10.0 ** rng.uniform(-8, 4, k - k // 2))])
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    want = (Q * np.maximum(lam, 0)) @ Q.T
    cone = dict(s=[k])
    m = capi.cone_rows(cone)
    kc = capi.make_cone(cone)
    c = lib._scs_init_cone(C.byref(kc), m)
    assert c
    # Proj_{K*}(v) with K* = K for the PSD cone
    x = _pack_psd(A)
    got = x.copy()
    assert lib._scs_proj_dual_cone(got.ctypes.data_as(T.fp), c, None, None) == 0
    G = _unpack_psd(got, k)
    scale = np.abs(A).max()
    assert np.abs(G - want).max() <= 1e-9 * scale, np.abs(G - want).max() / scale
    ev = np.linalg.eigvalsh(G)
    assert ev.min() >= -1e-9 * scale
    again = x.copy()                                   # second call: warm started from the carried eigenbasis
    assert lib._scs_proj_dual_cone(again.ctypes.data_as(T.fp), c, None, None) == 0
    assert np.abs(again - got).max() <= 1e-9 * scale
    fix = got.copy()                                   # a PSD matrix is its own projection
    assert lib._scs_proj_dual_cone(fix.ctypes.data_as(T.fp), c, None, None) == 0
    assert np.abs(fix - got).max() <= 1e-9 * scale
    lib._scs_finish_cone(c)


@pytest.mark.parametrize("cone", [dict(s=[73, 80, 92, 50]), dict(s=[88], cs=[40, 46])])
def test_lds_kernel_orders_73_to_92_warm_started_match_reference(cone):
    """Round 4: the LDS Jacobi kernel carries the eigenbasis for EVERY order it handles (73..92 keep T = A Vp in an HBM scratch;
    complex blocks of order 37..46 come through their real embedding of order 74..92).  A drifting sequence of projections -- cold,
    then warm started from the previous basis -- against the reference's LAPACK path at 1e-11 each time."""
    ref = _ref_lib()
    lib = _lib()
    Tr = ref._scs_types
    m = capi.cone_rows(cone)
    kr = capi.make_cone(cone, Tr)
    wr = ref._scs_init_cone(C.byref(kr), m)
    k = capi.make_cone(cone)
    c = lib._scs_init_cone(C.byref(k), m)
    assert c and wr
    rng = np.random.default_rng(11)
    x0 = rng.standard_normal(m)
    for rep in range(6):
        x0 = x0 + (0.3 if rep < 3 else 1e-3) * rng.standard_normal(m)   # large moves first, then the small ones warm starts are for
        got, want = x0.copy(), x0.copy()
        assert lib._scs_proj_dual_cone(got.ctypes.data_as(T.fp), c, None, None) == 0
        assert ref._scs_proj_dual_cone(want.ctypes.data_as(Tr.fp), wr, None, None) == 0
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert err <= 1e-11, (cone, rep, err)
    lib._scs_finish_cone(c)
    ref._scs_finish_cone(wr)


@pytest.mark.parametrize("cone", [dict(s=[100, 131, 300]), dict(s=[257], cs=[70])])
def test_fused_step_equals_the_two_launch_form(cone, monkeypatch):
    """Round 4: one launch per outer step of the blocked Jacobi iteration (k_bj_fused: the inner sweep of step k + 1 beside the
    update of step k, forming its subproblem from pre-update data with the update's own tile products).  The subproblem must have
    the bits the update writes, so the whole projection must equal the two-launch form's (SCS_AMD_PSD_FUSED=0, read at init) to
    rounding -- over a cold call and two warm-started ones, for blocks with different numbers of block columns in one launch."""
    lib = _lib()
    m = capi.cone_rows(cone)
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SCS_AMD_PSD_FUSED", fused)
        k = capi.make_cone(cone)
        c = lib._scs_init_cone(C.byref(k), m)
        assert c
        res = []
        x0 = np.random.default_rng(11).standard_normal(m)
        for rep in range(3):
            x = x0 + 0.05 * rep * np.random.default_rng(12 + rep).standard_normal(m)
            assert lib._scs_proj_dual_cone(x.ctypes.data_as(T.fp), c, None, None) == 0
            res.append(x)
        outs[fused] = res
        lib._scs_finish_cone(c)
    for a, b in zip(outs["1"], outs["0"]):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), np.abs(a - b).max()


@pytest.mark.parametrize("env", [dict(SCS_AMD_PSD_CROSS="0"), dict(SCS_AMD_PSD_CROSS="0", SCS_AMD_PSD_FUSED="0"), dict(SCS_AMD_PSD_BLOCKED="0")])
def test_measurement_forms_of_the_large_block_iteration_still_project(env, monkeypatch):
    """The forms kept for A/B measurements -- full 63-step subproblem sweeps (SCS_AMD_PSD_CROSS=0; through the fused step and as two
    launches) and round 2's single-column steps (SCS_AMD_PSD_BLOCKED=0) -- share the control record, the generic inner sweep and the
    update with the shipped form: they must still agree with it (different rotation orders: to 1e-10 of scale, not to rounding)."""
    lib = _lib()
    cone = dict(s=[100, 150])
    m = capi.cone_rows(cone)
    x0 = np.random.default_rng(21).standard_normal(m)

    def run():
        k = capi.make_cone(cone)
        c = lib._scs_init_cone(C.byref(k), m)
        assert c
        res = []
        for rep in range(2):
            x = x0 * (1.0 + 0.1 * rep)
            assert lib._scs_proj_dual_cone(x.ctypes.data_as(T.fp), c, None, None) == 0
            res.append(x)
        lib._scs_finish_cone(c)
        return res

    want = run()
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    got = run()
    for a, b in zip(got, want):
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), (env, np.abs(a - b).max())


@pytest.mark.parametrize("env", [dict(SCS_AMD_PSD_OFFSCAN="0"), dict(SCS_AMD_PSD_OFFSCAN="2"), dict(SCS_AMD_PSD_GRID="0"), dict(SCS_AMD_PSD_PROLOGUE="0"),
                                 dict(SCS_AMD_PSD_OFFSCAN="0", SCS_AMD_PSD_GRID="0", SCS_AMD_PSD_PROLOGUE="0")])
def test_round_6_forms_of_the_large_block_iteration_change_no_bit(env, monkeypatch):
    """Round 6 (profiles/r6_psd_big.md): the pass over the matrix that replaces the closing sweep (a sweep that rotates nothing changes
    nothing), the one-dimensional grid of the fused step (same jobs, another hand-out order) and the one-level prologue of its inner
    sweep (same entries, asked for at once) are schedules, not arithmetic: against rounds 4-5's forms (options psd_offscan = 0,
    psd_grid = 0, psd_prologue = 0) every projection -- cold and warm started, real and Hermitian blocks of
    several sizes in one cone -- must come out bit for bit the same."""
    lib = _lib()
    cone = dict(s=[100, 150, 97], cs=[60])
    m = capi.cone_rows(cone)
    x0 = np.random.default_rng(33).standard_normal(m)

    def run():
        k = capi.make_cone(cone)
        c = lib._scs_init_cone(C.byref(k), m)
        assert c
        res = []
        for rep in range(4):
            x = x0 + 0.02 * rep * np.random.default_rng(40 + rep).standard_normal(m)
            assert lib._scs_proj_dual_cone(x.ctypes.data_as(T.fp), c, None, None) == 0
            res.append(x)
        lib._scs_finish_cone(c)
        return res

    want = run()
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    got = run()
    for a, b in zip(got, want):
        assert np.array_equal(a, b), (env, np.abs(a - b).max())


# ---- round 6: the pipelined kernel's NB = 2 instantiation (every order 51..72) on hardware ----------------
# psd_blocks_per_lane(npairs, 448) is 1 up to order 50 and 2 for every order 51..72 (from 51 the V-row groups no longer fit the 448
# update lanes, from 59 neither do the blocks); 72 is the largest order whose three matrices fit the LDS.  None of these had been
# compared with the reference on a GPU: the CPU restatement (tests/test_psd_step_host.py) loops lanes and cannot see the DPP lane
# shifts, the register ties, the LDS ping-pong or the real barrier.
def _bind_options(lib):
    lib.scs_amd_set_option.restype = C.c_int
    lib.scs_amd_set_option.argtypes = [C.c_char_p, C.c_char_p]
    return lib


_NB2_CONES = [dict(s=[50, 51, 52, 58, 59, 64, 71, 72]),   # both sides of the NB 1 -> 2 switch, both reasons for NB = 2, odd orders (padding index), the LDS boundary
              dict(cs=[26, 30, 36]),                      # Hermitian blocks through their real embeddings of order 52 / 60 / 72
              dict(s=[72, 50, 7]),                        # one launch, blocks of different NB (the switch is per workgroup)
              dict(l=4, q=[9], s=[63, 65, 2, 1], cs=[33, 1])]


@pytest.mark.parametrize("cone", _NB2_CONES)
def test_pipelined_psd_kernel_orders_51_to_72_match_reference_cold_and_warm(cone):
    """Against the live reference's _scs_proj_dual_cone (LAPACK dsyevr / zheevr, src/cones.c:999-1155) at 1e-11: a cold projection,
    then a 10-step drifting sequence warm started from the carried eigenbasis (large moves first, then the small ones warm starts are
    for), then an unrelated input.  The same sequence through the two-phase step (option psd_pipe = 0) must agree with the pipelined
    one to rounding: same rotations in the same order, the look-ahead's diagonal entries differ by O(eps |a|) only."""
    ref = _ref_lib()
    lib = _bind_options(_lib())
    Tr = ref._scs_types
    m = capi.cone_rows(cone)
    kr = capi.make_cone(cone, Tr)
    wr = ref._scs_init_cone(C.byref(kr), m)
    assert wr
    rng = np.random.default_rng(61)
    seq = []
    x0 = rng.standard_normal(m)
    for rep in range(12):
        if rep == 0:
            pass
        elif rep == 11:
            x0 = 5.0 * rng.standard_normal(m)                          # unrelated: the carried basis is useless, not harmful
        else:
            x0 = x0 + (0.3 if rep < 4 else 1e-3) * rng.standard_normal(m)
        seq.append(x0.copy())
    want = []
    for x in seq:
        w = x.copy()
        assert ref._scs_proj_dual_cone(w.ctypes.data_as(Tr.fp), wr, None, None) == 0
        want.append(w)
    ref._scs_finish_cone(wr)
    got = {}
    try:
        for pipe in ("1", "0", "2"):   # 2 (round 6): the signal form of the pipelined step
            assert lib.scs_amd_set_option(b"psd_pipe", pipe.encode()) == 0   # read by _scs_init_cone's device workspace
            k = capi.make_cone(cone)
            c = lib._scs_init_cone(C.byref(k), m)
            assert c
            res = []
            for rep, x in enumerate(seq):
                g = x.copy()
                assert lib._scs_proj_dual_cone(g.ctypes.data_as(T.fp), c, None, None) == 0
                err = np.abs(g - want[rep]).max() / max(1.0, np.abs(want[rep]).max())
                assert err <= 1e-11, (cone, pipe, rep, err)
                assert np.abs(g - x).max() > 1e-3
                res.append(g)
            got[pipe] = res
            lib._scs_finish_cone(c)
    finally:
        lib.scs_amd_set_option(b"psd_pipe", None)
    for rep, (a, b) in enumerate(zip(got["1"], got["0"])):
        assert np.abs(a - b).max() <= 2e-12 * max(1.0, np.abs(b).max()), (cone, rep, np.abs(a - b).max())
    for rep, (a, b) in enumerate(zip(got["2"], got["0"])):   # the signal form applies the two-phase step's own rotations
        assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(b).max()), (cone, rep, np.abs(a - b).max())


def test_pipelined_psd_kernel_every_order_from_2_to_72_against_numpy():
    """Every order the pipelined instantiation handles, one launch each (so every order also appears as the launch's LARGEST block,
    which sizes the LDS and the stride of the carried basis), cold and once warm, against numpy's eigh at 1e-11."""
    from scs_amd import problems
    lib = capi.load("libscsamd.so")
    worst = 0.0
    for k0 in range(2, 73):
        cone = dict(s=[k0, max(1, k0 - 1)])
        m = capi.cone_rows(cone)
        k = capi.make_cone(cone)
        w = lib.scs_amd_cone_init(C.byref(k), m, None)
        assert w
        x0 = np.random.default_rng(1000 + k0).standard_normal(m)
        for rep in range(2):
            x = x0 * (1.0 + 0.01 * rep)
            want = problems.proj_dual_cone_np(x, cone)
            got = x.copy()
            assert lib.scs_amd_cone_proj_dual(w, got.ctypes.data_as(T.fp), None) == 0
            err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
            assert err <= 1e-11, (k0, rep, err)
            worst = max(worst, err)
        lib.scs_amd_cone_finish(w)
    assert worst <= 1e-11


def test_pipelined_psd_kernel_fp32_orders_64_and_72():
    """The -DSFLOAT build of the same instantiation (its threshold is the fp32 rounding floor of the rotations, cones.hip): orders
    64 and 72 (+ the NB switch at 50 / 51) against the float64 numpy projection at fp32 accuracy, cold and over a warm sequence."""
    from scs_amd import problems
    lib = capi.load("libscsamd_f32.so")
    T32 = capi.T32
    cone = dict(s=[64, 72, 50, 51])
    m = capi.cone_rows(cone)
    k = capi.make_cone(cone, T32)
    w = lib.scs_amd_cone_init(C.byref(k), m, None)
    assert w
    rng = np.random.default_rng(17)
    v = rng.standard_normal(m)
    for rep in range(6):
        v = v + (0.3 if rep < 3 else 1e-3) * rng.standard_normal(m)
        x = v.astype(np.float32)
        want = problems.proj_dual_cone_np(x.astype(np.float64), cone)
        assert lib.scs_amd_cone_proj_dual(w, x.ctypes.data_as(T32.fp), None) == 0
        err = np.abs(x - want).max() / max(1.0, np.abs(want).max())
        assert err <= 2e-4, (rep, err)
        assert np.abs(x - v).max() > 1e-2
    lib.scs_amd_cone_finish(w)


def test_blocked_psd_iteration_fp32_orders_beyond_the_lds_path():
    """The -DSFLOAT build of the blocked iteration (psd_big.h: plain-loop products, the same scan / norm / prologue / grid as the fp64
    build): orders 100, 130 and 97 in one cone against the float64 numpy projection at fp32 accuracy, cold and over a warm-started
    sequence."""
    from scs_amd import problems
    lib = capi.load("libscsamd_f32.so")
    T32 = capi.T32
    cone = dict(s=[100, 130, 97])
    m = capi.cone_rows(cone)
    k = capi.make_cone(cone, T32)
    w = lib.scs_amd_cone_init(C.byref(k), m, None)
    assert w
    rng = np.random.default_rng(23)
    v = rng.standard_normal(m)
    for rep in range(5):
        v = v + (0.3 if rep < 2 else 1e-3) * rng.standard_normal(m)
        x = v.astype(np.float32)
        want = problems.proj_dual_cone_np(x.astype(np.float64), cone)
        assert lib.scs_amd_cone_proj_dual(w, x.ctypes.data_as(T32.fp), None) == 0
        err = np.abs(x - want).max() / max(1.0, np.abs(want).max())
        assert err <= 5e-4, (rep, err)
        assert np.abs(x - v).max() > 1e-2
    lib.scs_amd_cone_finish(w)
