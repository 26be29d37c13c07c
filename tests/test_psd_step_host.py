"""The LDS Jacobi step of k_psd_jacobi (scs_amd/csrc/psd_lds_step.h) compiled for the HOST (tests/native/host_check_psd_step.cpp: the
same lane-level functions, lanes looped one at a time, a barrier = the end of a loop).  Round 5 pipelines the step -- a look-ahead wave
forms step s+1's rotations from step s's tables while step s is applied to the other copy of A, one barrier per step -- and this pins
the schedule without a GPU:

 * the circle-method rule the look-ahead relies on (pair i of step s+1 takes its players from pairs i+1 / i-1 of step s) holds for
   every order the LDS kernel handles;
 * the pipelined iteration produces the SAME rotations as the two-phase step of rounds 2-4 (eigenvalues and eigenvectors equal to
   rounding, equal sweep / step counts);
 * both agree with numpy's eigh (the LAPACK route of reference src/cones.c:999-1067)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = os.path.join(str(tmp_path_factory.mktemp("psdstep")), "libpsdstep.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "native", "host_check_psd_step.cpp")])
    l = C.CDLL(so)
    l.psd_check_eig.argtypes = [DP, C.c_int, C.c_int, DP, DP, C.POINTER(C.c_long)]
    return l


def _eig(lib, a, pipelined):
    k = a.shape[0]
    a = np.ascontiguousarray(a, dtype=np.float64)
    ev, vec, cnt = np.zeros(k), np.zeros((k, k)), (C.c_long * 4)()
    rc = lib.psd_check_eig(a.ctypes.data_as(DP), k, pipelined, ev.ctypes.data_as(DP), vec.ctypes.data_as(DP), cnt)
    return rc, ev, vec, list(cnt)


def _sym(k, seed, spread=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((k, k))
    a = a + a.T
    if spread:  # wide eigenvalue spread
        q, _ = np.linalg.qr(rng.standard_normal((k, k)))
        a = (q * np.logspace(-spread, spread, k)) @ q.T
        a = 0.5 * (a + a.T)
    return a


@pytest.mark.parametrize("k", [2, 3, 4, 5, 7, 8, 16, 31, 49, 50, 64, 71, 72])
def test_pipelined_step_equals_the_two_phase_step_and_numpy(lib, k):
    for seed, spread in ((k, 0), (100 + k, 4)):
        a = _sym(k, seed, spread)
        rc0, e0, v0, c0 = _eig(lib, a, 0)
        rc1, e1, v1, c1 = _eig(lib, a, 1)
        assert rc0 == 0 and rc1 == 0
        assert c1[3] == 0, "look-ahead read a pair that does not hold its player"
        assert abs(c0[0] - c1[0]) <= 1 and c0[2] // max(c0[0], 1) == c1[2] // max(c1[0], 1), (c0, c1)  # same schedule; sweeps within one
        scale = np.abs(e0).max()
        # (the look-ahead forms the rotated diagonal entries by the classical a_pp -/+ t a_xy instead of restating the update's block
        # product: equal to rounding, so the two iterations agree to rounding, not bit for bit)
        # (eigenvectors of a spectrum spread over 1e-4 .. 1e4 are conditioned like eps |A| / gap ~ 1e-8: compared where they are well posed)
        assert np.abs(e0 - e1).max() <= 1e-12 * scale and (spread or np.abs(v0 - v1).max() <= 1e-11)
        w = np.linalg.eigvalsh(a)
        assert np.abs(np.sort(e1) - w).max() <= 1e-12 * scale
        assert np.abs(v1 @ v1.T - np.eye(k)).max() <= 1e-12
        assert np.abs((v1 * e1) @ v1.T - a).max() <= 1e-12 * scale


def test_warm_started_matrix_skips_steps_identically(lib):
    """a nearly diagonal matrix (what the warm start hands over): most steps rotate nothing and are skipped -- the pipelined form must
    skip the same ones (the look-ahead through identity tables is exact)"""
    k = 50
    rng = np.random.default_rng(3)
    a = np.diag(rng.standard_normal(k) * 10)
    for (i, j) in ((3, 17), (20, 21), (0, 49), (8, 30)):
        a[i, j] = a[j, i] = 1e-6 * rng.standard_normal()
    _, e0, v0, c0 = _eig(lib, a, 0)
    _, e1, v1, c1 = _eig(lib, a, 1)
    assert c0[:3] == c1[:3] and c1[1] < c1[2] and c1[3] == 0, (c0, c1)
    assert np.abs(e0 - e1).max() <= 1e-13 * np.abs(e0).max() and np.abs(v0 - v1).max() <= 1e-12


@pytest.mark.parametrize("k", [2, 3, 4, 5, 6, 7, 8, 16, 31, 49, 50, 51, 64, 65, 66, 71, 72])
def test_signal_form_is_the_two_phase_iteration_bit_for_bit(lib, k):
    """Round 6: in the signal form of the pipelined step the update lanes own the 2 npairs PRIORITY blocks first and the rotation wave
    reads (a_pq, a_pp, a_qq) of its next pair from the updated copy.  Pinned here: the enumeration of the blocks (priority first) still
    covers every block exactly once (the result equals the two-phase iteration BIT FOR BIT: same rotations from the same stored values),
    and every entry the rotation wave reads lies in a priority block of the step being applied (counter 3 == 0), for every parity of
    the order, both blocks-per-lane instantiations and the padded odd orders."""
    for seed, spread in ((k, 0), (200 + k, 4)):
        a = _sym(k, seed, spread)
        rc0, e0, v0, c0 = _eig(lib, a, 0)
        rc2, e2, v2, c2 = _eig(lib, a, 2)
        assert rc0 == 0 and rc2 == 0
        assert c2[3] == 0, "the rotation wave read an entry outside the priority blocks"
        assert c0[:3] == c2[:3], (c0, c2)
        assert np.array_equal(e0, e2) and np.array_equal(v0, v2)
