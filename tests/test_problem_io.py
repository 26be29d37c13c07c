"""Problem-file I/O (scs_amd/csrc/problem_io.cpp) against the reference's src/rw.c in
both directions (CPU only): what we write the reference reads back identically, what the
reference writes we read back identically, and the reference's own fixture parses."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle
from scs_amd import capi, problems

T = capi.T64
pytestmark = pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")


def _bind(amd, ref):
    PD, PK, PS = C.POINTER(T.ScsData), C.POINTER(T.ScsCone), C.POINTER(T.ScsSettings)
    amd.scs_amd_write_data.restype = C.c_int
    amd.scs_amd_write_data.argtypes = [PD, PK, PS, C.c_char_p]
    amd.scs_amd_read_data.restype = C.c_int
    amd.scs_amd_read_data.argtypes = [C.c_char_p, C.POINTER(PD), C.POINTER(PK), C.POINTER(PS)]
    amd.scs_amd_free_data.argtypes = [PD, PK, PS]
    ref._scs_read_data.restype = C.c_int
    ref._scs_read_data.argtypes = [C.c_char_p, C.POINTER(PD), C.POINTER(PK), C.POINTER(PS)]
    ref._scs_write_data.restype = None
    ref._scs_write_data.argtypes = [PD, PK, PS]
    return PD, PK, PS


def _as_dict(d, k, s):
    d, k, s = d.contents, k.contents, s.contents
    A = d.A.contents
    nnz = A.p[d.n]
    out = dict(m=d.m, n=d.n, b=np.ctypeslib.as_array(d.b, (d.m,)).copy(), c=np.ctypeslib.as_array(d.c, (d.n,)).copy(),
               Ap=np.ctypeslib.as_array(A.p, (d.n + 1,)).copy(), Ai=np.ctypeslib.as_array(A.i, (nnz,)).copy(),
               Ax=np.ctypeslib.as_array(A.x, (nnz,)).copy(), hasP=bool(d.P),
               z=k.z, l=k.l, bsize=k.bsize, qsize=k.qsize, ssize=k.ssize, cssize=k.cssize, ep=k.ep, ed=k.ed, psize=k.psize)
    if k.bsize > 1:
        out["bl"] = np.ctypeslib.as_array(k.bl, (k.bsize - 1,)).copy()
        out["bu"] = np.ctypeslib.as_array(k.bu, (k.bsize - 1,)).copy()
    for name, cnt in (("q", k.qsize), ("s", k.ssize), ("cs", k.cssize)):
        if cnt:
            out[name] = np.ctypeslib.as_array(getattr(k, name), (cnt,)).copy()
    if k.psize:
        out["p"] = np.ctypeslib.as_array(k.p, (k.psize,)).copy()
    if d.P:
        P = d.P.contents
        out["Px"] = np.ctypeslib.as_array(P.x, (P.p[d.n],)).copy()
    for f in ("normalize", "scale", "rho_x", "max_iters", "eps_abs", "eps_rel", "eps_infeas", "alpha", "verbose",
              "acceleration_lookback", "acceleration_interval", "acceleration_type_1", "acceleration_regularization",
              "acceleration_relaxation", "adaptive_scale", "time_limit_secs"):
        out["stg_" + f] = getattr(s, f)
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for key in a:
        if isinstance(a[key], np.ndarray):
            assert np.array_equal(a[key], b[key]), key
        else:
            assert a[key] == b[key], key


def _problem():
    import scipy.sparse as sp
    cone = dict(z=2, l=3, bl=[-1.0, -2.0], bu=[1.0, 3.0], q=[3, 4], s=[2, 3], cs=[2], ep=1, ed=2, p=[0.3, -0.6])
    m = capi.cone_rows(cone)
    rng = np.random.default_rng(0)
    A = sp.random(m, 7, density=0.3, random_state=1, format="csc")
    P = sp.random(7, 7, density=0.3, random_state=2, format="csc")
    return capi.Problem(A, rng.standard_normal(m), rng.standard_normal(7), cone, P=(P + P.T).tocsc())


def test_we_write_reference_reads_and_back(tmp_path):
    amd, ref = capi.load("libscsamd.so"), pyoracle.load_ref()
    PD, PK, PS = _bind(amd, ref)
    prob = _problem()
    st = capi.default_settings(amd, max_iters=777, eps_abs=3e-5, acceleration_type_1=0, time_limit_secs=12.5, scale=0.7)
    f1 = str(tmp_path / "ours.bin").encode()
    assert amd.scs_amd_write_data(C.byref(prob.data), C.byref(prob.k), C.byref(st), f1) == 0
    d, k, s = PD(), PK(), PS()
    assert ref._scs_read_data(f1, C.byref(d), C.byref(k), C.byref(s)) == 0          # reference reads ours
    d2, k2, s2 = PD(), PK(), PS()
    assert amd.scs_amd_read_data(f1, C.byref(d2), C.byref(k2), C.byref(s2)) == 0     # and so do we
    a, b = _as_dict(d, k, s), _as_dict(d2, k2, s2)
    _same(a, b)
    assert a["stg_max_iters"] == 777 and a["stg_time_limit_secs"] == 12.5 and a["cssize"] == 1 and a["hasP"]
    assert np.array_equal(a["Ax"], prob.Ax) and np.array_equal(a["b"], prob.b)
    # reference writes what it read; the two files are byte-identical
    f2 = str(tmp_path / "theirs.bin").encode()
    s.contents.write_data_filename = f2
    ref._scs_write_data(d, k, s)
    assert open(f1, "rb").read() == open(f2, "rb").read()
    amd.scs_amd_free_data(d2, k2, s2)


def test_reference_fixture_parses(tmp_path):
    amd, ref = capi.load("libscsamd.so"), pyoracle.load_ref()
    PD, PK, PS = _bind(amd, ref)
    path = os.path.join(pyoracle.REF_DIR, "test", "problems", "random_prob")
    if not os.path.exists(path):
        pytest.skip("reference fixture not staged (make -C oracle conform)")
    d, k, s = PD(), PK(), PS()
    d2, k2, s2 = PD(), PK(), PS()
    assert ref._scs_read_data(path.encode(), C.byref(d), C.byref(k), C.byref(s)) == 0
    assert amd.scs_amd_read_data(path.encode(), C.byref(d2), C.byref(k2), C.byref(s2)) == 0
    _same(_as_dict(d, k, s), _as_dict(d2, k2, s2))
    amd.scs_amd_free_data(d2, k2, s2)


def test_missing_file_is_an_error_not_a_crash():
    amd, ref = capi.load("libscsamd.so"), pyoracle.load_ref()
    PD, PK, PS = _bind(amd, ref)
    d, k, s = PD(), PK(), PS()
    assert amd.scs_amd_read_data(b"/nonexistent/file", C.byref(d), C.byref(k), C.byref(s)) == -1
    assert not d and not k and not s
