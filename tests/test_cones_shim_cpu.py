"""libscsamd_cones.so = boundary B1' in drop-in form: the nine `_scs_*` symbols of the
reference's internal cone interface (include/cones.h:80-90).  Everything except the
projection itself is host code and is checked here without a GPU; the projection goes
through the reference's own test-suite on the GPU (oracle/_ref/run_tests_amd_cones,
profiles/r1_conformance_b1p_*.log) and tests/test_cones_shim_gpu.py."""
import ctypes as C
import subprocess

import numpy as np
import pytest

from scs_amd import capi

T = capi.T64
NINE = {"_scs_init_cone", "_scs_proj_dual_cone", "_scs_finish_cone", "_scs_set_r_y", "_scs_enforce_cone_boundaries",
        "_scs_validate_cones", "_scs_get_cone_header", "_scs_deep_copy_cone", "_scs_free_cone"}


@pytest.fixture(scope="module")
def lib():
    l = C.CDLL(capi.lib_path("libscsamd_cones.so"))
    l._scs_init_cone.restype = C.c_void_p
    l._scs_init_cone.argtypes = [C.POINTER(T.ScsCone), C.c_int]
    l._scs_finish_cone.argtypes = [C.c_void_p]
    l._scs_set_r_y.argtypes = [C.c_void_p, C.c_double, T.fp]
    l._scs_validate_cones.argtypes = [C.POINTER(T.ScsData), C.POINTER(T.ScsCone)]
    l._scs_get_cone_header.restype = C.c_void_p
    l._scs_get_cone_header.argtypes = [C.POINTER(T.ScsCone)]
    l._scs_deep_copy_cone.argtypes = [C.POINTER(T.ScsCone), C.POINTER(T.ScsCone)]
    return l


def test_exports_exactly_the_cone_interface():
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.lib_path("libscsamd_cones.so")], text=True)
    exp = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert NINE <= exp
    assert all(s in NINE or s.startswith("scs_amd_") for s in exp), exp
    hdr = open(capi.lib_path("libscsamd_cones.so").replace("scs_amd/lib/libscsamd_cones.so", "include/scs_amd.h")).read()
    for s in NINE:
        assert s + "(" in hdr, s     # every one is declared in include/scs_amd.h


def _data(m):
    d = T.ScsData()
    d.m, d.n = m, 1
    return d


def test_validate_cones_follows_the_reference_cases(lib):
    # the cases of the reference's test/problems/test_validation.h:130-232
    ok = capi.make_cone(dict(bu=[1.0, np.inf], bl=[-np.inf, -1.0]))
    assert lib._scs_validate_cones(C.byref(_data(3)), C.byref(ok)) == 0          # one-sided infinite bounds pass
    bad = capi.make_cone(dict(bu=[1.0], bl=[np.nan]))
    assert lib._scs_validate_cones(C.byref(_data(2)), C.byref(bad)) < 0          # NaN bound
    bad = capi.make_cone(dict(bu=[np.inf], bl=[np.inf]))
    assert lib._scs_validate_cones(C.byref(_data(2)), C.byref(bad)) < 0          # +inf lower bound
    bad = capi.make_cone(dict(bu=[1.0], bl=[5.0]))
    assert lib._scs_validate_cones(C.byref(_data(2)), C.byref(bad)) < 0          # bl > bu
    bad = capi.make_cone(dict(p=[2.0]))
    assert lib._scs_validate_cones(C.byref(_data(3)), C.byref(bad)) < 0          # power out of [-1, 1]
    bad = capi.make_cone(dict(l=3))
    assert lib._scs_validate_cones(C.byref(_data(4)), C.byref(bad)) < 0          # rows do not add up
    bad = capi.make_cone(dict(l=3))
    bad.z = -1
    assert lib._scs_validate_cones(C.byref(_data(2)), C.byref(bad)) < 0
    bad = capi.make_cone(dict(l=3))
    bad.qsize = 1                                                                # size array missing
    assert lib._scs_validate_cones(C.byref(_data(3)), C.byref(bad)) < 0


def test_set_r_y_boundaries_header_and_copies(lib):
    cone = dict(z=2, l=3, bu=[1.0], bl=[-1.0], q=[3, 4], s=[2], ep=1, p=[0.5])
    m = capi.cone_rows(cone)
    k = capi.make_cone(cone)
    c = lib._scs_init_cone(C.byref(k), m)
    assert c
    r = np.zeros(m)
    lib._scs_set_r_y(c, 0.1, r.ctypes.data_as(T.fp))                             # src/cones.c:349-363
    assert np.allclose(r[:2], 1.0 / (1000 * 0.1)) and np.allclose(r[2:], 1.0 / 0.1)
    # enforce_cone_boundaries with f = max: first block (z + l + box) untouched, one value per cone after
    CB = C.CFUNCTYPE(C.c_double, T.fp, C.c_int)
    fmax = CB(lambda p, n: max(p[i] for i in range(n)))
    lib._scs_enforce_cone_boundaries.argtypes = [C.c_void_p, T.fp, CB]
    v = np.arange(m, dtype=np.float64)
    want = v.copy()
    pos = 2 + 3 + 2
    for ln in [3, 4, 3, 3, 3]:
        want[pos:pos + ln] = want[pos:pos + ln].max()
        pos += ln
    lib._scs_enforce_cone_boundaries(c, v.ctypes.data_as(T.fp), fmax)
    assert np.array_equal(v, want)
    lib._scs_finish_cone(c)
    assert not lib._scs_init_cone(C.byref(k), m + 1)                             # inconsistent with m
    h = lib._scs_get_cone_header(C.byref(k))
    txt = C.string_at(h).decode()
    assert "soc vars: 7, qsize: 2" in txt and "psd vars: 3" in txt and "exp vars: 3" in txt
    C.CDLL(None).free(C.c_void_p(h))
    # deep copy owns its arrays; free_cone releases them and the struct (calloc'd like the reference's)
    libc = C.CDLL(None)
    libc.calloc.restype = C.c_void_p
    dst = C.cast(libc.calloc(1, C.sizeof(T.ScsCone)), C.POINTER(T.ScsCone))
    assert lib._scs_deep_copy_cone(dst, C.byref(k)) == 1
    d = dst.contents
    assert (d.z, d.l, d.bsize, d.qsize, d.ssize, d.ep, d.psize) == (2, 3, 2, 2, 1, 1, 1)
    assert [d.q[i] for i in range(2)] == [3, 4] and d.s[0] == 2 and d.bu[0] == 1.0 and d.p[0] == 0.5
    assert C.addressof(d.q.contents) != C.addressof(k.q.contents)
    lib._scs_free_cone.argtypes = [C.POINTER(T.ScsCone)]
    lib._scs_free_cone(dst)
