"""The exponential / power cone arithmetic of scs_amd/csrc/cones_exp_pow.h, compiled for the HOST
(tests/native/host_check_exp_pow.cpp: the same functions, one lane at a time) and pinned without a GPU

 (a) to the reference's golden vectors (tests/golden/cones.npz: 40 + 35 exponential, 48 power cones) at 1e-12;
 (b) to the live reference (`_scs_proj_dual_cone` of oracle/_ref, src/exp_cone.c:373-441, src/cones.c:1290-1335) on
     thousands of random triples at several scales, near the cone surface, and in ill-conditioned extremes -- there the
     reference's own root search gives up and returns one of its closed-form candidates; the check is then the
     definition of a projection: our point lies in the cone and is at least as close to the input.

The GPU kernel (`k_exp_pow`) is compared with the reference through the same goldens in tests/test_cones_shim_gpu.py /
tests/test_golden_gpu.py and with this host build in tests/test_cones_exp_pow_gpu.py."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from scs_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
DP = C.POINTER(C.c_double)


def build_host_check(tmpdir, f32=False):
    so = os.path.join(str(tmpdir), "libxpcheck32.so" if f32 else "libxpcheck.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC"] + (["-DXP_CHECK_FLOAT"] if f32 else []) +
                          ["-o", so, os.path.join(HERE, "native", "host_check_exp_pow.cpp")])
    lib = C.CDLL(so)
    lib.xp_check_project_exp.argtypes = [C.POINTER(C.c_float) if f32 else DP, C.c_int]
    lib.xp_check_project_pow.argtypes = [C.POINTER(C.c_float), C.c_float] if f32 else [DP, C.c_double]
    lib._ftype = (np.float32, C.POINTER(C.c_float)) if f32 else (np.float64, DP)
    return lib


@pytest.fixture(scope="module")
def xp(tmp_path_factory):
    return build_host_check(tmp_path_factory.mktemp("xpcheck"))


def proj_cone_host(lib, v, ep, ed, p):
    """Proj_K of every triple of v (K = exp | dual exp | power with a < 0 meaning the dual power cone)."""
    ft, ptr = lib._ftype
    out = np.array(v, dtype=ft)
    for c in range(ep + ed + len(p)):
        t = np.ascontiguousarray(out[3 * c:3 * c + 3])
        if c < ep + ed:
            lib.xp_check_project_exp(t.ctypes.data_as(ptr), int(c >= ep))
        else:
            a = p[c - ep - ed]
            if a >= 0:
                lib.xp_check_project_pow(t.ctypes.data_as(ptr), a)
            else:  # Moreau, src/cones.c:1427-1441
                w = -t.copy()
                lib.xp_check_project_pow(w.ctypes.data_as(ptr), -a)
                t = t + w
        out[3 * c:3 * c + 3] = t
    return out


def proj_dual_host(lib, x, ep, ed, p):
    return x + proj_cone_host(lib, -x, ep, ed, p)  # src/cones.c:1552-1596


def ref_proj_dual(x, cone):
    from oracle import pyoracle
    ref = pyoracle.load_ref()
    k = capi.make_cone(cone)
    c = ref._scs_init_cone(C.byref(k), capi.cone_rows(cone))
    y = x.copy()
    assert ref._scs_proj_dual_cone(y.ctypes.data_as(capi.T64.fp), c, None, None) == 0
    ref._scs_finish_cone(c)
    return y


def test_fp32_host_build_matches_fp32_reference(tmp_path):
    """-DSFLOAT arithmetic: the reference's own fp32 build is the yardstick (its power-cone Newton iteration, followed
    step for step here, loses its footing in fp32 on some triples -- both builds then return the same far-off point; the
    reference documents SFLOAT as "currently broken", docs/src/api/compile_flags.rst:25-28)."""
    from oracle import pyoracle
    if not pyoracle.ref_available("libscsindir_ref_f32.so"):
        pytest.skip("oracle/_ref f32 flavour not built")
    ref = pyoracle.load_ref("libscsindir_ref_f32.so")
    T = capi.T32
    xp32 = build_host_check(tmp_path, f32=True)
    rng = np.random.default_rng(3)
    ep, ed, p = 400, 300, [0.5, -0.3, 0.8] * 50
    cone = dict(ep=ep, ed=ed, p=p)
    x = rng.standard_normal(3 * (ep + ed + len(p))).astype(np.float32)
    k = capi.make_cone(cone, T)
    c = ref._scs_init_cone(C.byref(k), capi.cone_rows(cone))
    want = x.copy()
    assert ref._scs_proj_dual_cone(want.ctypes.data_as(T.fp), c, None, None) == 0
    ref._scs_finish_cone(c)
    got = proj_dual_host(xp32, x, ep, ed, p)
    assert np.abs(got - want)[:3 * (ep + ed)].max() <= 5e-4   # two different fp32 root searches on the same F
    assert np.abs(got - want)[3 * (ep + ed):].max() <= 2e-5   # power cones: the same iteration, step for step


def test_host_build_matches_reference_golden_vectors(xp):
    g = np.load(os.path.join(G, "cones.npz"))
    meta = json.load(open(os.path.join(G, "cones_meta.json")))
    for name in ("exp", "pow"):
        cone = meta[name]
        x, want = np.array(g[f"{name}_eucl_x"]), g[f"{name}_eucl_y"]
        got = proj_dual_host(xp, x, cone.get("ep", 0), cone.get("ed", 0), cone.get("p", []))
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), name


def _need_ref():
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")


def _rel_err(got, want, x):
    """per-cone error relative to the size of the cone's input (x + Proj(-x) cancels: its own size says nothing)"""
    return np.abs(got - want).reshape(-1, 3).max(1) / np.maximum(1.0, np.abs(x).reshape(-1, 3).max(1))


@pytest.mark.parametrize("scale", [1.0, 1e-3, 30.0])
def test_random_triples_match_live_reference(xp, scale):
    _need_ref()
    rng = np.random.default_rng(int(scale * 1000) + 1)
    n = 3000
    x = rng.standard_normal(3 * 2 * n) * scale
    want = ref_proj_dual(x, dict(ep=n, ed=n))
    got = proj_dual_host(xp, x, n, n, [])
    e = _rel_err(got, want, x)
    assert e.max() <= 1e-9, e.max()              # near-degenerate rays (t -> 0+): both root searches sit in rounding noise
    assert (e <= 1e-12).mean() >= 0.99
    p = list(rng.uniform(0.05, 0.95, n) * np.where(rng.random(n) < 0.5, 1.0, -1.0))
    x = rng.standard_normal(3 * n) * scale
    want = ref_proj_dual(x, dict(p=p))
    got = proj_dual_host(xp, x, 0, 0, p)
    assert _rel_err(got, want, x).max() <= 1e-12


@pytest.mark.parametrize("eps", [1e-3, 1e-6, 1e-9, 1e-12, -1e-9])
def test_points_near_the_cone_surface_match_live_reference(xp, eps):
    """inside (eps < 0) and just outside the surface, where the reference keeps its closed-form candidate below 1e-8"""
    _need_ref()
    rng = np.random.default_rng(5)
    n = 1500
    w = np.exp(rng.uniform(-2, 2, n))
    u = w * rng.uniform(-8, 8, n)
    t = w * np.exp(u / w) * (1 - eps * rng.uniform(0.1, 1, n))
    v = np.stack([u, w, t], 1).ravel()
    for cone, (ep, ed) in ((dict(ep=n), (n, 0)), (dict(ed=n), (0, n))):
        want = ref_proj_dual(-v, cone)
        got = proj_dual_host(xp, -v, ep, ed, [])
        assert _rel_err(got, want, v).max() <= 1e-12


def _in_exp_cone(p, tol):
    u, w, t = p[:, 0], p[:, 1], p[:, 2]
    scale = np.maximum(1.0, np.abs(p).max(1))
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        surf = np.where(w > 0, w * np.exp(np.where(w > 0, u / np.where(w > 0, w, 1.0), 0.0)), 0.0)
    face = (np.abs(w) <= tol * scale) & (u <= tol * scale) & (t >= -tol * scale)
    return face | ((w > 0) & (surf <= t + tol * np.maximum(scale, np.abs(t))))


def test_ill_conditioned_triples_are_projections_at_least_as_good_as_the_reference(xp):
    """dynamic ranges up to e^12 between the rows (u / w up to ~60, t up to 1e13): where the two differ, ours must be
    the better projection -- in the cone, and no farther from the input"""
    _need_ref()
    rng = np.random.default_rng(9)
    n = 4000
    v = rng.standard_normal(3 * n) * np.exp(rng.uniform(-6, 6, 3 * n))
    ref_p = ref_proj_dual(-v, dict(ep=n)) + v    # Proj_K(v) as the reference sees it
    our_p = proj_cone_host(xp, v, n, 0, [])
    V, R, O = v.reshape(-1, 3), ref_p.reshape(-1, 3), our_p.reshape(-1, 3)
    assert _in_exp_cone(O, 1e-9).all()
    d_ref, d_our = np.linalg.norm(R - V, axis=1), np.linalg.norm(O - V, axis=1)
    assert np.all(d_our <= d_ref * (1 + 1e-9) + 1e-9 * np.maximum(1.0, np.abs(V).max(1)))
    close = np.abs(O - R).max(1) <= 1e-9 * np.maximum(1.0, np.abs(V).max(1))
    assert close.mean() >= 0.95


@pytest.mark.parametrize("f32", [False, True])
def test_root_search_trip_histogram(tmp_path, f32):
    """the bracketed Newton search had no 'Newton converged' exit, so a lane whose step rounded to zero
    bisected ~50 more trips and -- the loop being wave-uniform -- nearly every wave paid them (per searched triple, 20000
    random triples: median 11, p90 56, p99 60 evaluations of F).  With the exit the tail is gone: the assertion pins it."""
    lib = build_host_check(tmp_path, f32=f32)
    lib.xp_check_take_eval_count.restype = C.c_long
    ft, ptr = lib._ftype
    rs = np.random.RandomState(7)
    counts = []
    for _ in range(20000):
        t = np.ascontiguousarray(rs.randn(3) * 10.0 ** rs.uniform(-1, 1), dtype=ft)
        lib.xp_check_take_eval_count()
        lib.xp_check_project_exp(t.ctypes.data_as(ptr), int(rs.rand() < 0.5))
        c = lib.xp_check_take_eval_count()
        if c:
            counts.append(c)
    counts = np.array(counts)
    med, p90, p99, mx = np.percentile(counts, [50, 90, 99, 100])
    print(json.dumps(dict(f32=f32, searched=int(len(counts)), median=float(med), p90=float(p90), p99=float(p99), max=float(mx))))
    assert len(counts) > 5000
    assert p99 <= 30 and mx <= 40, (med, p90, p99, mx)  # before the fix (fp64): p90 56, p99 60, max 71; after: p99 15, max 22 (fp32: 25, 36)
