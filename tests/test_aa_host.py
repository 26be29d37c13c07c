"""Host Anderson acceleration (scs_amd/csrc/aa_host.cpp) vs the reference's src/aa.c
on the same fixed-point iteration (CPU only; the AA code is pure host code)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle
from scs_amd import capi

pytestmark = pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
dp = C.POINTER(C.c_double)


def _libs():
    amd = capi.load("libscsamd.so")
    amd.scs_amd_aa_init.restype = C.c_void_p
    amd.scs_amd_aa_init.argtypes = [C.c_int] * 4 + [C.c_double] * 4 + [C.c_int]
    amd.scs_amd_aa_apply.restype = C.c_double
    amd.scs_amd_aa_apply.argtypes = [dp, dp, C.c_void_p]
    amd.scs_amd_aa_safeguard.restype = C.c_int
    amd.scs_amd_aa_safeguard.argtypes = [dp, dp, C.c_void_p]
    amd.scs_amd_aa_finish.argtypes = [C.c_void_p]
    amd.scs_amd_aa_reset.argtypes = [C.c_void_p]
    ref = pyoracle.load_ref()
    ref.aa_init.restype = C.c_void_p
    ref.aa_init.argtypes = [C.c_int] * 4 + [C.c_double] * 4 + [C.c_int, C.c_int]
    ref.aa_apply.restype = C.c_double
    ref.aa_apply.argtypes = [dp, dp, C.c_void_p]
    ref.aa_safeguard.restype = C.c_int
    ref.aa_safeguard.argtypes = [dp, dp, C.c_void_p]
    ref.aa_finish.argtypes = [C.c_void_p]
    ref.aa_reset.argtypes = [C.c_void_p]
    return amd, ref


def _p(a):
    return a.ctypes.data_as(dp)


@pytest.mark.parametrize("type1,reg,relax,mem", [(1, 1e-8, 1.0, 10), (0, 1e-12, 1.0, 5), (1, 1e-8, 1.3, 6), (1, -1e-6, 1.0, 4)])
def test_aa_matches_reference_on_a_contraction(type1, reg, relax, mem):
    amd, ref = _libs()
    rng = np.random.default_rng(5)
    dim = 300
    Q, _ = np.linalg.qr(rng.standard_normal((dim, dim)))
    M = Q @ np.diag(rng.uniform(0.3, 0.98, dim)) @ Q.T   # contraction
    c = rng.standard_normal(dim)
    F = lambda v: M @ v + c + 0.05 * np.maximum(v, 0)     # mildly nonlinear map

    def run(lib, init, apply, safeguard, finish, extra):
        a = init(dim, mem, mem, type1, reg, relax, 1.0, 1e10, 5, *extra)
        x = np.zeros(dim)
        x_prev = x.copy()
        norms, traj = [], []
        for i in range(60):
            if i > 0:
                norms.append(apply(_p(x), _p(x_prev), a))
            x_prev = x.copy()
            x = F(x)
            rej = safeguard(_p(x), _p(x_prev), a)
            traj.append((rej, x.copy()))
        finish(a)
        return norms, traj

    na, ta = run(amd, amd.scs_amd_aa_init, amd.scs_amd_aa_apply, amd.scs_amd_aa_safeguard, amd.scs_amd_aa_finish, ())
    nr, tr = run(ref, ref.aa_init, ref.aa_apply, ref.aa_safeguard, ref.aa_finish, (0,))
    assert [a[0] for a in ta] == [r[0] for r in tr]                      # same safeguard decisions
    assert np.all(np.sign(na) == np.sign(nr))                            # same accept / reject pattern
    for (ra, xa), (rr, xr) in zip(ta, tr):
        assert np.abs(xa - xr).max() <= 1e-7 * max(1.0, np.abs(xr).max())
    assert any(v > 0 for v in na)  # some steps were accepted
