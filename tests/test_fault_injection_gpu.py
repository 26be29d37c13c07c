"""The failure convention under injected HIP failures.

`scs_amd_test_fail_at(k)` makes the k-th HIP runtime call the library checks report hipErrorOutOfMemory
(scs_amd/csrc/common.h: every allocation, copy, synchronisation and post-launch poll goes through HIP_CHECK).  What the
reference's conventions then demand (src/scs.c:361-371 failure(), :1381-1384, :1092-1096; include/linsys.h:25-71):

 * scs_init -> NULL, scs_init_lin_sys_work -> NULL, nothing leaked on the device;
 * scs_solve -> SCS_FAILED (-4), status "failure", iter -1, NaN-filled solution, the SIGINT handler restored;
 * scs_solve_lin_sys -> non-zero;
 * the library is usable afterwards and gives the same answer as before.
"""
import ctypes as C
import os
import signal

import numpy as np
import pytest

from scs_amd import capi, problems

pytestmark = pytest.mark.gpu


def _free_bytes(lib):
    """hipMemGetInfo through the library itself (torch's own HIP runtime cannot be initialised in a process in which the
    library's is already live: "No HIP GPUs are available")"""
    v = lib.scs_amd_device_free_bytes()
    assert v >= 0
    return v


@pytest.fixture(scope="module")
def prob():
    pr = problems.random_socp(30000, 60000, 10, seed=11)
    return capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])


def _checked_calls(lib, fn):
    """how many checked HIP calls `fn` makes: arm far ahead, run, read what is left"""
    big = 10 ** 12
    lib.scs_amd_test_fail_at(big)
    fn()
    return big - lib.scs_amd_test_fail_at(0)


def test_scs_init_returns_null_and_leaks_nothing(prob):
    lib = capi.load("libscsamd.so")
    st = capi.default_settings(lib, verbose=0, acceleration_lookback=0)
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))  # warm: HIP context, streams, code objects
    assert w
    lib.scs_finish(w)
    held = []
    total = _checked_calls(lib, lambda: held.append(lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))))  # scs_init alone
    lib.scs_finish(held.pop())
    assert total > 50
    base = _free_bytes(lib)
    # early (device selection / first allocations), middle (equilibration, matrix layouts), late (cone tables, g solve)
    for k in sorted({1, 2, 3, 7, total // 4, total // 2, (3 * total) // 4, total - 5}):
        lib.scs_amd_test_fail_at(k)
        w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
        left = lib.scs_amd_test_fail_at(0)
        assert left == 0, (k, left)   # the injected failure was consumed inside scs_init
        assert not w, k               # NULL, as src/scs.c:1092-1096 / :1279-1283
        assert abs(_free_bytes(lib) - base) <= 8 << 20, (k, base - _free_bytes(lib))  # partial state freed (HIP caches a few MB of its own)
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    assert w
    lib.scs_finish(w)


def test_scs_solve_fails_with_nan_solution_and_recovers(prob):
    lib = capi.load("libscsamd.so")
    good = capi.solve(lib, prob, verbose=0, acceleration_lookback=0, max_iters=60)
    T = lib._scs_types
    st = capi.default_settings(lib, verbose=0, acceleration_lookback=0, max_iters=60)
    before = signal.getsignal(signal.SIGINT)
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    assert w
    x, y, s = np.zeros(prob.n), np.zeros(prob.m), np.zeros(prob.m)
    sol = T.ScsSolution(x.ctypes.data_as(T.fp), y.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp))
    info = T.ScsInfo()
    total = _checked_calls(lib, lambda: lib.scs_solve(w, C.byref(sol), C.byref(info), 0))
    assert info.status_val in (1, 2) and total > 100
    # (a re-solve on a used workspace starts from the adapted scale, so its call count can differ a little: stay clear of the end)
    for k in (1, 5, total // 3, total // 2, (3 * total) // 4):
        x[:] = y[:] = s[:] = 0
        lib.scs_amd_test_fail_at(k)
        rc = lib.scs_solve(w, C.byref(sol), C.byref(info), 0)
        assert lib.scs_amd_test_fail_at(0) == 0, k
        assert rc == -4 and info.status_val == -4 and info.status == b"failure" and info.iter == -1, (k, rc, info.status)  # SCS_FAILED
        assert np.all(np.isnan(x)) and np.all(np.isnan(y)) and np.all(np.isnan(s)), k
        assert np.isnan(info.pobj) and np.isnan(info.res_pri)
        assert signal.getsignal(signal.SIGINT) == before  # ctrl-c handler restored on the failure path too (src/scs.c:369)
    # the same workspace still solves, bit-identically to the solve before the failures
    rc = lib.scs_solve(w, C.byref(sol), C.byref(info), 0)
    assert rc == info.status_val and info.status_val in (1, 2)
    lib.scs_finish(w)
    again = capi.solve(lib, prob, verbose=0, acceleration_lookback=0, max_iters=60)
    assert np.array_equal(again["x"], good["x"]) and again["info"]["pobj"] == good["info"]["pobj"]
    # a delivered SIGINT still reaches Python's own handler
    with pytest.raises(KeyboardInterrupt):
        os.kill(os.getpid(), signal.SIGINT)
        import time
        time.sleep(1.0)


def test_linsys_plugin_boundary_under_failures(prob):
    """B1: init -> NULL (include/linsys.h:25-33), solve -> non-zero (the reference then aborts with SCS_FAILED,
    src/scs.c:1381-1384), update -> negative (src/scs.c:1221-1223)."""
    from tests import probgen
    lib = capi.load("libscsamd_linsys.so")
    T = lib._scs_types
    n, m = prob.n, prob.m
    dr = probgen.diag_r(n, m, z=m // 10)
    fp = lambda a: a.ctypes.data_as(T.fp)
    w = lib.scs_init_lin_sys_work(C.byref(prob.matA), None, fp(dr))
    assert w
    rng = np.random.default_rng(1)
    b = rng.uniform(-1, 1, n + m)
    want = b.copy()
    assert lib.scs_solve_lin_sys(w, fp(want), None, 1e-9) == 0
    lib.scs_free_lin_sys_work(w)
    base = _free_bytes(lib)
    held = []
    total = _checked_calls(lib, lambda: held.append(lib.scs_init_lin_sys_work(C.byref(prob.matA), None, fp(dr))))
    lib.scs_free_lin_sys_work(held.pop())
    assert total > 10
    for k in (1, 2, total // 2, total - 2):
        lib.scs_amd_test_fail_at(k)
        w = lib.scs_init_lin_sys_work(C.byref(prob.matA), None, fp(dr))
        assert lib.scs_amd_test_fail_at(0) == 0
        assert not w, k
        assert abs(_free_bytes(lib) - base) <= 8 << 20, k
    w = lib.scs_init_lin_sys_work(C.byref(prob.matA), None, fp(dr))
    assert w
    o = b.copy()
    total = _checked_calls(lib, lambda: lib.scs_solve_lin_sys(w, fp(o), None, 1e-9))
    assert total >= 4
    # (the number of checked calls of a solve depends on how the previous one sized its batches: stay at the front, where the
    # upload, the first launches' poll and the first control-block read-back are)
    for k in (1, 2, 3):
        o = b.copy()
        lib.scs_amd_test_fail_at(k)
        rc = lib.scs_solve_lin_sys(w, fp(o), None, 1e-9)
        assert lib.scs_amd_test_fail_at(0) == 0
        assert rc != 0, k
    lib.scs_amd_test_fail_at(1)
    assert lib.scs_update_lin_sys_diag_r(w, fp(dr)) < 0
    assert lib.scs_amd_test_fail_at(0) == 0
    # still usable: same answer as before the failures
    assert lib.scs_update_lin_sys_diag_r(w, fp(dr)) == 0
    o = b.copy()
    assert lib.scs_solve_lin_sys(w, fp(o), None, 1e-9) == 0
    assert np.array_equal(o, want)
    lib.scs_free_lin_sys_work(w)
