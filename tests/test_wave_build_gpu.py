"""The wave-owned-rows layout built ON THE DEVICE (scs_amd/csrc/spmv_wave_build.h, round 5) against the host builder of rounds
2-4 (WaveRowsDev::fill_host -- the oracle): SCS_AMD_WR_BUILD=verify builds both inside scs_init and compares every byte of the
packed words, the values and the distinct-line count; scs_init fails (NULL) on any difference.  Then the solves: a workspace whose
layouts were built on the device returns bit-identical iterates to one built on the host."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from scs_amd import capi, problems

pytestmark = pytest.mark.gpu


def _layout_info(lib, prob, **over):
    st = capi.default_settings(lib, verbose=0, **over)
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    assert w, "scs_init returned NULL (SCS_AMD_WR_BUILD=verify: the device-built layout differs from the host builder's)"
    out = (C.c_double * 6)()
    lib.scs_amd_get_layout_info(w, out)
    lib.scs_finish(w)
    return list(out)


CASES = [
    # (n, m, col_nnz, band, lockstep env, note)
    (30000, 60000, 10, None, "0", "uniformly random, plain kernel's layout"),
    (30000, 60000, 10, None, "1", "uniformly random, lockstep layout (quarter-window chunk order)"),
    (30000, 60000, 10, 512, "1", "banded (gathers share lines), lockstep layout"),
    (2000, 9000, 37, None, "1", "short units with ragged last chunks"),
    (50000, 52000, 3, None, "0", "very sparse rows"),
]


@pytest.mark.parametrize("n,m,col_nnz,band,lockstep,note", CASES)
@pytest.mark.parametrize("libname", ["libscsamd.so", "libscsamd_f32.so"])
def test_device_layout_equals_host_layout(monkeypatch, libname, n, m, col_nnz, band, lockstep, note):
    lib = capi.load(libname)
    T = lib._scs_types
    pr = problems.random_socp(n, m, col_nnz, seed=n + col_nnz, band=band)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"], T=T)
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")      # these sizes are below the library's own threshold
    monkeypatch.setenv("SCS_AMD_WR_LOCKSTEP", lockstep)
    monkeypatch.setenv("SCS_AMD_WR_BUILD", "verify")
    monkeypatch.setenv("SCS_AMD_TRANSPOSE", "verify")  # the pattern transpose of the device equilibration against the host loop
    info = _layout_info(lib, prob)
    assert info[0] == 1 and info[3] == 1, info        # both orientations carry the layout ...
    assert info[1] == 1 and info[4] == 1, info        # ... and the device built it (verify compared it with the host's)
    assert 0 < info[2] <= 1.0 + 1e-9 and 0 < info[5] <= 1.0 + 1e-9


@pytest.mark.parametrize("lockstep", ["0", "1"])
@pytest.mark.parametrize("libname", ["libscsamd.so", "libscsamd_f32.so"])
def test_device_layout_equals_host_layout_under_the_chain_and_home_numbering(monkeypatch, libname, lockstep):
    """Round 6: under the chain + home numbering (forced on at this size) a fifth of a unit's entries sit in the few column buckets
    around the unit's home -- units with very uneven buckets, which the uniformly random and the banded cases above do not have.
    verify mode compares every byte of the two builders' output; the numbering must really be in force (lines per entry below 0.9).
    (Moving such heavy buckets to the front of the unit was built and measured: slower, profiles/r6_chain_home.md -- not kept.)"""
    lib = capi.load(libname)
    T = lib._scs_types
    pr = problems.random_socp(30000, 60000, 10, seed=77)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"], T=T)
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")
    monkeypatch.setenv("SCS_AMD_REORDER", "1")
    monkeypatch.setenv("SCS_AMD_WR_LOCKSTEP", lockstep)
    monkeypatch.setenv("SCS_AMD_WR_BUILD", "verify")
    monkeypatch.setenv("SCS_AMD_TRANSPOSE", "verify")
    info = _layout_info(lib, prob)
    assert info[0] == 1 and info[3] == 1 and info[1] == 1 and info[4] == 1, info
    assert info[2] < 0.9 and info[5] < 0.9, info


def test_rows_of_a_few_thousand_entries_stay_on_the_device(monkeypatch):
    """a row of 3000 entries: the device transpose sorts it with one workgroup (bitonic in LDS), and its unit still fits the device
    layout builder"""
    lib = capi.load("libscsamd.so")
    rng = np.random.default_rng(1)
    n, m = 12000, 30000
    pr = problems.random_socp(n, m, 10, seed=6)
    A = sp.csc_matrix(pr["A"]).tolil()
    A[11, rng.choice(n, 3000, replace=False)] = rng.standard_normal(3000) * 0.01
    A[12, rng.choice(n, 40, replace=False)] = rng.standard_normal(40) * 0.01
    A = sp.csc_matrix(A)
    A.sort_indices()
    prob = capi.Problem(A, pr["b"], pr["c"], pr["cone"])
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")
    monkeypatch.setenv("SCS_AMD_WR_BUILD", "verify")
    monkeypatch.setenv("SCS_AMD_TRANSPOSE", "verify")
    for lock in ("0", "1"):
        monkeypatch.setenv("SCS_AMD_WR_LOCKSTEP", lock)
        info = _layout_info(lib, prob)
        assert info[0] == 1 and info[1] == 1 and info[3] == 1 and info[4] == 1, info


def test_unit_too_long_for_lds_takes_the_host_builder(monkeypatch):
    """a row with more than 8192 entries is a unit of its own that does not fit the device builder's LDS sort: the host builder takes
    the whole matrix (still verified end to end by the solve below matching the CSR-stream path)"""
    lib = capi.load("libscsamd.so")
    rng = np.random.default_rng(0)
    n, m = 12000, 30000
    pr = problems.random_socp(n, m, 8, seed=5)
    A = sp.csc_matrix(pr["A"]).tolil()
    cols = rng.choice(n, 9000, replace=False)
    A[7, cols] = rng.standard_normal(9000) * 0.01   # one dense-ish row of A (a long unit of CSR(A))
    A = sp.csc_matrix(A)
    A.sort_indices()
    prob = capi.Problem(A, pr["b"], pr["c"], pr["cone"])
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")
    monkeypatch.setenv("SCS_AMD_WR_BUILD", "verify")
    monkeypatch.setenv("SCS_AMD_TRANSPOSE", "verify")  # (a 9000-entry row also exceeds the device transpose's in-LDS sort: host loop)
    info = _layout_info(lib, prob)
    assert info[0] == 1 and info[1] == 0, info        # CSR(A): host builder
    assert info[3] == 1 and info[4] == 1, info        # CSR(A'): columns of A are short -> device


def test_solves_are_bit_identical_whichever_side_built_the_layout(monkeypatch):
    lib = capi.load("libscsamd.so")
    pr = problems.random_socp(40000, 80000, 10, seed=9)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    monkeypatch.setenv("SCS_AMD_WAVEROWS", "1")
    out = {}
    for lock in ("0", "1"):
        monkeypatch.setenv("SCS_AMD_WR_LOCKSTEP", lock)
        for mode in ("host", "dev"):
            monkeypatch.setenv("SCS_AMD_WR_BUILD", mode)
            monkeypatch.setenv("SCS_AMD_TRANSPOSE", mode)
            out[lock, mode] = capi.solve(lib, prob, verbose=0, acceleration_lookback=0, max_iters=40)
        a, b = out[lock, "host"], out[lock, "dev"]
        assert a["info"]["iter"] == b["info"]["iter"] == 40
        for v in ("x", "y", "s"):
            assert np.array_equal(a[v], b[v]), (lock, v)
