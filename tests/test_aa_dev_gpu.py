"""Device-resident Anderson acceleration (scs_amd/csrc/aa_dev.hip) vs the reference's
src/aa.c (oracle/_ref) and vs the host restatement, on the same fixed-point iteration.
The device path reorders the O(dim) sums, so agreement is to rounding amplified by the
(regularised) least-squares solve: 1e-6 relative on the iterates, identical accept /
reject and safeguard decisions."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle
from scs_amd import capi

pytestmark = pytest.mark.gpu
dp = C.POINTER(C.c_double)


def _amd():
    amd = capi.load("libscsamd.so")
    for pre in ("scs_amd_aa_", "scs_amd_aa_dev_"):
        getattr(amd, pre + "init").restype = C.c_void_p
        getattr(amd, pre + "init").argtypes = [C.c_int] * 4 + [C.c_double] * 4 + [C.c_int]
        getattr(amd, pre + "apply").restype = C.c_double
        getattr(amd, pre + "apply").argtypes = [dp, dp, C.c_void_p]
        getattr(amd, pre + "safeguard").restype = C.c_int
        getattr(amd, pre + "safeguard").argtypes = [dp, dp, C.c_void_p]
        getattr(amd, pre + "finish").argtypes = [C.c_void_p]
        getattr(amd, pre + "reset").argtypes = [C.c_void_p]
    return amd


def _ref():
    ref = pyoracle.load_ref()
    ref.aa_init.restype = C.c_void_p
    ref.aa_init.argtypes = [C.c_int] * 4 + [C.c_double] * 4 + [C.c_int, C.c_int]
    ref.aa_apply.restype = C.c_double
    ref.aa_apply.argtypes = [dp, dp, C.c_void_p]
    ref.aa_safeguard.restype = C.c_int
    ref.aa_safeguard.argtypes = [dp, dp, C.c_void_p]
    ref.aa_finish.argtypes = [C.c_void_p]
    return ref


def _p(a):
    return a.ctypes.data_as(dp)


def _map(dim, seed):
    """A cheap contraction with a mild nonlinearity: banded mixing, no dense matrix."""
    rng = np.random.default_rng(seed)
    d0 = rng.uniform(0.3, 0.95, dim)
    d1 = rng.uniform(-0.02, 0.02, dim)
    c = rng.standard_normal(dim)
    return lambda v: d0 * v + d1 * np.roll(v, 1) + c + 0.03 * np.maximum(v, 0)


def _run(init, apply, safeguard, finish, extra, F, dim, mem, type1, reg, relax, iters=50):
    a = init(dim, mem, mem, type1, reg, relax, 1.0, 1e10, 5, *extra)
    assert a
    x = np.zeros(dim)
    x_prev = x.copy()
    norms, traj = [], []
    for i in range(iters):
        if i > 0:
            norms.append(apply(_p(x), _p(x_prev), a))
        x_prev = x.copy()
        x = F(x)
        rej = safeguard(_p(x), _p(x_prev), a)
        traj.append((rej, x.copy()))
    finish(a)
    return norms, traj


CASES = [
    (1, 1e-8, 1.0, 10, 70001),   # SCS defaults (type-I, lookback 10); 21 panel columns = 2 batches
    (0, 1e-12, 1.0, 5, 5003),    # type-II
    (1, 1e-8, 1.3, 6, 40000),    # relaxation
    (1, -1e-6, 1.0, 4, 300),     # pinned regularisation, a single workgroup
    (0, 1e-10, 1.0, 20, 9000),   # lookback > one batch of pivot candidates
]


@pytest.mark.parametrize("type1,reg,relax,mem,dim", CASES)
def test_device_aa_matches_host_restatement(type1, reg, relax, mem, dim):
    amd = _amd()
    F = _map(dim, 7)
    nd, td = _run(amd.scs_amd_aa_dev_init, amd.scs_amd_aa_dev_apply, amd.scs_amd_aa_dev_safeguard,
                  amd.scs_amd_aa_dev_finish, (), F, dim, mem, type1, reg, relax)
    nh, th = _run(amd.scs_amd_aa_init, amd.scs_amd_aa_apply, amd.scs_amd_aa_safeguard,
                  amd.scs_amd_aa_finish, (), F, dim, mem, type1, reg, relax)
    assert [a[0] for a in td] == [b[0] for b in th]
    assert np.all(np.sign(nd) == np.sign(nh))
    assert any(v > 0 for v in nd)
    np.testing.assert_allclose(nd, nh, rtol=1e-5, atol=1e-9)
    for (_, xd), (_, xh) in zip(td, th):
        assert np.abs(xd - xh).max() <= 1e-6 * max(1.0, np.abs(xh).max())


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("type1,reg,relax,mem,dim", CASES[:3])
def test_device_aa_matches_reference(type1, reg, relax, mem, dim):
    amd, ref = _amd(), _ref()
    F = _map(dim, 11)
    nd, td = _run(amd.scs_amd_aa_dev_init, amd.scs_amd_aa_dev_apply, amd.scs_amd_aa_dev_safeguard,
                  amd.scs_amd_aa_dev_finish, (), F, dim, mem, type1, reg, relax)
    nr, tr = _run(ref.aa_init, ref.aa_apply, ref.aa_safeguard, ref.aa_finish, (0,), F, dim, mem, type1, reg, relax)
    assert [a[0] for a in td] == [b[0] for b in tr]
    assert np.all(np.sign(nd) == np.sign(nr))
    for (_, xd), (_, xr) in zip(td, tr):
        assert np.abs(xd - xr).max() <= 1e-6 * max(1.0, np.abs(xr).max())


def test_device_aa_reset_and_rejects_bad_parameters():
    amd = _amd()
    assert not amd.scs_amd_aa_dev_init(100, 5, 5, 1, 1e-8, 3.0, 1.0, 1e10, 5)   # relaxation out of range
    a = amd.scs_amd_aa_dev_init(1000, 5, 5, 1, 1e-8, 1.0, 1.0, 1e10, 5)
    F = _map(1000, 3)
    x = np.zeros(1000)
    for i in range(8):
        xp = x.copy()
        x = F(x)
        amd.scs_amd_aa_dev_apply(_p(x), _p(xp), a)
    amd.scs_amd_aa_dev_reset(a)
    xp = x.copy()
    x1 = F(x)
    keep = x1.copy()
    assert amd.scs_amd_aa_dev_apply(_p(x1), _p(xp), a) == 0.0   # first call after a reset only seeds
    np.testing.assert_array_equal(x1, keep)
    amd.scs_amd_aa_dev_finish(a)
