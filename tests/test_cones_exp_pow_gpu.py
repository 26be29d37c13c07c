"""`k_exp_pow` (scs_amd/csrc/cones_exp_pow.h) on the GPU against the host build of the same arithmetic
(tests/native/host_check_exp_pow.cpp, itself pinned to the reference in tests/test_exp_pow_host.py) and against the
live reference: thousands of cones, cone counts that are not multiples of the 256-cone tile, all three cone kinds
meeting inside one tile and inside one wave, several scales, points on / near the surface."""
import ctypes as C

import numpy as np
import pytest

from scs_amd import capi
from tests.test_exp_pow_host import build_host_check, proj_dual_host, ref_proj_dual, _rel_err

pytestmark = pytest.mark.gpu
T = capi.T64


def _gpu_proj_dual(lib, x, cone):
    k = capi.make_cone(cone, T)
    w = lib.scs_amd_cone_init(C.byref(k), capi.cone_rows(cone), None)
    assert w
    y = x.copy()
    assert lib.scs_amd_cone_proj_dual(w, y.ctypes.data_as(T.fp), None) == 0
    lib.scs_amd_cone_finish(w)
    return y


@pytest.mark.parametrize("ep,ed,npow,scale", [(1, 0, 0, 1.0), (3, 2, 4, 1.0), (300, 211, 130, 1.0), (5000, 4099, 3001, 1.0),
                                               (700, 700, 700, 1e-3), (700, 700, 700, 30.0)])
def test_kernel_matches_host_build_and_reference(tmp_path, ep, ed, npow, scale):
    lib = capi.load("libscsamd.so")
    xp = build_host_check(tmp_path)
    rng = np.random.default_rng(ep + 7 * ed + 13 * npow)
    p = list(rng.uniform(0.05, 0.95, npow) * np.where(rng.random(npow) < 0.5, 1.0, -1.0))
    cone = dict(ep=ep, ed=ed, p=p)
    x = rng.standard_normal(3 * (ep + ed + npow)) * scale
    # a few exactly-on-the-surface, inside and trivially-signed triples among the random ones
    if ep >= 300:
        x[0:3] = [-1.0, -2.0, -np.exp(0.5) * 2.0]          # -x on the surface
        x[3:6] = [1.0, 1.0, -5.0]                          # -x = (-1, -1, 5): (u, w) <= 0
        x[6:9] = [-0.3, -1.0, -4.0]                        # -x strictly inside K
        x[9:12] = [0.0, 0.0, 0.0]
    got = _gpu_proj_dual(lib, x, cone)
    host = proj_dual_host(xp, x, ep, ed, p)
    eh = _rel_err(got, host, x)                            # device exp / pow differ from glibc's by ulps; where the root
    assert eh.max() <= 1e-10 and (eh <= 1e-12).mean() >= 0.99, eh.max()   # sits in F's rounding noise that shows at 1e-11
    from oracle import pyoracle
    if pyoracle.ref_available():
        want = ref_proj_dual(x, cone)
        e = _rel_err(got, want, x)
        assert e.max() <= 1e-9 and (e <= 1e-12).mean() >= 0.99, e.max()


def test_fp32_build_of_the_kernel(tmp_path):
    """the -DSFLOAT kernel against the fp32 host build of the same arithmetic and the reference's own fp32 build"""
    lib = capi.load("libscsamd_f32.so")
    T32 = capi.T32
    rng = np.random.default_rng(3)
    ep, ed, p = 400, 300, [0.5, -0.3, 0.8] * 50
    cone = dict(ep=ep, ed=ed, p=p)
    k = capi.make_cone(cone, T32)
    w = lib.scs_amd_cone_init(C.byref(k), capi.cone_rows(cone), None)
    assert w
    x = rng.standard_normal(3 * (ep + ed + len(p))).astype(np.float32)
    y = x.copy()
    assert lib.scs_amd_cone_proj_dual(w, y.ctypes.data_as(T32.fp), None) == 0
    lib.scs_amd_cone_finish(w)
    host = proj_dual_host(build_host_check(tmp_path, f32=True), x, ep, ed, p)
    assert np.abs(y - host)[:3 * (ep + ed)].max() <= 5e-4      # device expf vs glibc's inside an fp32 root search
    from oracle import pyoracle
    if pyoracle.ref_available("libscsindir_ref_f32.so"):
        ref = pyoracle.load_ref("libscsindir_ref_f32.so")
        kk = capi.make_cone(cone, T32)
        c = ref._scs_init_cone(C.byref(kk), capi.cone_rows(cone))
        want = x.copy()
        assert ref._scs_proj_dual_cone(want.ctypes.data_as(T32.fp), c, None, None) == 0
        ref._scs_finish_cone(c)
        assert np.abs(y - want)[:3 * (ep + ed)].max() <= 1e-3
        # power cones: the reference's Newton iteration in fp32 arithmetic; a device powf that differs by an ulp can stop it one
        # step earlier or later -- all but a handful of cones agree closely, every cone to the iteration's own 1e-9 / fp32 slack
        d = np.abs(y - want)[3 * (ep + ed):].reshape(-1, 3).max(1)
        assert (d <= 1e-4).mean() >= 0.95
