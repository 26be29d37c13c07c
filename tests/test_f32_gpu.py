"""BASELINE configs[4] in miniature: the -DSFLOAT build (libscsamd_f32.so) against the
reference's SFLOAT=1 build on the same inputs; tolerance 1e-3 as the config states (the
reference documents SFLOAT as "currently broken", docs/src/api/compile_flags.rst:25-28)."""
import numpy as np
import pytest

from scs_amd import capi, problems

pytestmark = pytest.mark.gpu


def test_fp32_solve_matches_fp32_reference_loosely():
    from oracle import pyoracle
    if not pyoracle.ref_available("libscsindir_ref_f32.so"):
        pytest.skip("oracle/_ref f32 flavour not built")
    ref = pyoracle.load_ref("libscsindir_ref_f32.so")
    amd = capi.load("libscsamd_f32.so")
    pr = problems.random_socp(400, 1200, 8, seed=4, dtype=np.float32)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"], T=capi.T32)
    kw = dict(verbose=0, acceleration_lookback=0, eps_abs=1e-3, eps_rel=1e-3, max_iters=5000)
    ra, rr = capi.solve(amd, prob, **kw), capi.solve(ref, prob, **kw)
    assert ra["info"]["status_val"] in (1, 2) and rr["info"]["status_val"] in (1, 2)
    scale = max(1.0, abs(rr["info"]["pobj"]))
    assert abs(ra["info"]["pobj"] - rr["info"]["pobj"]) <= 2e-2 * scale
    popt = float(pr["c"].astype(np.float64) @ pr["x_opt"])
    assert abs(ra["info"]["pobj"] - popt) <= 2e-2 * max(1.0, abs(popt))


def test_fp32_cone_projections_including_large_psd_and_large_box():
    """The -DSFLOAT build of the cone kernels: PSD blocks in LDS (40), beyond the LDS path (110: chip-wide Jacobi
    steps with the fp32 scalar products of psd_big.h), a second-order cone and a 20 000-row box (chip-wide Newton
    steps), against the float64 numpy projection at fp32 accuracy; twice (the second call is warm-started)."""
    import ctypes as C
    lib = capi.load("libscsamd_f32.so")
    T = capi.T32
    nb = 20000
    rng = np.random.default_rng(3)
    cone = dict(l=7, bu=rng.uniform(0.5, 2.0, nb), bl=-rng.uniform(0.5, 2.0, nb), q=[300], s=[40, 110])
    m = capi.cone_rows(cone)
    k = capi.make_cone(cone, T)
    w = lib.scs_amd_cone_init(C.byref(k), m, None)
    assert w
    for rep in range(2):
        v = np.random.default_rng(10 + rep).standard_normal(m)
        x = v.astype(np.float32)
        assert lib.scs_amd_cone_proj_dual(w, x.ctypes.data_as(T.fp), None) == 0
        want = problems.proj_dual_cone_np(x.astype(np.float64), cone)
        err = np.abs(x - want).max() / max(1.0, np.abs(want).max())
        assert err <= 2e-4, (rep, err)
        assert np.abs(x - v).max() > 1e-2
    lib.scs_amd_cone_finish(w)
