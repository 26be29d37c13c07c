"""The NATIVE row-sharded PCG (scs_amd/csrc/shard_native.cpp): device-controlled loop, one all-reduce of an
n-vector per CG iteration enqueued on the solver's stream, no host read-back per iteration.

 * N = 2 and N = 3 through the in-process test double of the collective (ranks = host threads sharing the one GPU of a test box;
   RCCL refuses two ranks on one device): the split algebra -- R_x / N per slab, summed preconditioner, r_x counted once, slab-local
   back-substitution -- against the unsplit solve (<= 1e-8) and the reference backend (<= 1e-7);
 * N = 1 through RCCL's C API (ncclAllReduce on the solver's stream): same answer as the unsplit solve, the collective timed with
   HIP events at the headline's n = 1e6 (8 MB per CG iteration) -- the figure DESIGN.md's xGMI model stands next to."""
import ctypes as C
import threading

import numpy as np
import pytest

from scs_amd import capi, shard
from tests import probgen

pytestmark = pytest.mark.gpu


def _unsplit(A, dr, b, s, tol):
    lib = capi.load("libscsamd_linsys.so")
    T = lib._scs_types
    m, n = A.shape
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
    w = lib.scs_init_lin_sys_work(C.byref(prob.matA), None, dr.ctypes.data_as(T.fp))
    assert w
    o = b.copy()
    assert lib.scs_solve_lin_sys(w, o.ctypes.data_as(T.fp), None if s is None else s.ctypes.data_as(T.fp), tol) == 0
    st = T.ScsAmdStats()
    lib.scs_amd_linsys_get_stats(w, C.byref(st))
    lib.scs_free_lin_sys_work(w)
    return o, int(st.cg_iters)


@pytest.mark.parametrize("world", [2, 3])
def test_thread_ranks_on_one_gpu_match_the_unsplit_solve(world):
    lib = capi.load("libscsamd.so")
    n, m = 6000, 15001
    A = probgen.random_csc(m, n, 8, seed=5)
    dr = probgen.diag_r(n, m, z=m // 10)
    rng = np.random.default_rng(3)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n) * 0.1
    want, its = _unsplit(A, dr, b, s, 1e-11)
    want2, _ = _unsplit(A, dr, 2 * b, None, 1e-11)
    dr2 = probgen.diag_r(n, m, z=m // 10, scale=0.7)     # what a scale update of the ADMM loop hands to the backend
    want3, _ = _unsplit(A, dr2, b, s, 1e-11)
    group = lib.scs_amd_shard_group_create(world)
    assert group
    res, err = [None] * world, []

    def rank_main(r):
        try:
            S = shard.NativeShardedLinSys(A, dr, world=world, rank=r, group=group, lib=lib)
            x, y = S.solve(b, s, tol=1e-11)
            x2, y2 = S.solve(2 * b, None, tol=1e-11)  # a second, cold-started solve on the same workspaces
            st = S.stats()
            # scs_update_lin_sys_diag_r of the split system (collective: the preconditioner is re-summed over the ranks)
            T = lib._scs_types
            loc = np.concatenate([dr2[:n] / world, dr2[n + S.r0:n + S.r1]]).astype(T.np_float)
            assert lib.scs_amd_shard_update_diag_r(S.h, loc.ctypes.data_as(T.fp)) == 0
            x3, y3 = S.solve(b, s, tol=1e-11)
            res[r] = (x, y, x2, y2, st, (S.r0, S.r1), x3, y3)
            S.close()
        except Exception as e:  # (a rank that dies would leave the others waiting in the barrier: report and let the test time out)
            err.append(repr(e))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not err and all(r is not None for r in res), err
    lib.scs_amd_shard_group_free(group)
    sc = np.abs(want).max()
    y = np.concatenate([r[1] for r in res])
    assert [r[5] for r in res] == [shard.slab(m, world, q) for q in range(world)]
    for r in res:
        assert np.array_equal(r[0], res[0][0])                      # the replicated part: identical bits on every rank
        assert np.abs(r[0] - want[:n]).max() <= 1e-8 * sc
        assert np.abs(r[2] - want2[:n]).max() <= 1e-8 * np.abs(want2).max()
        st = r[4]
        assert st["solves"] == 2 and abs(st["cg_iters"] - res[0][4]["cg_iters"]) == 0
        assert st["allreduces"] >= st["cg_iters"]                   # one n-vector all-reduce per CG iteration (+ setup)
    assert np.abs(y - want[n:]).max() <= 1e-8 * sc
    y3 = np.concatenate([r[7] for r in res])
    assert np.abs(res[0][6] - want3[:n]).max() <= 1e-8 * np.abs(want3).max() and np.abs(y3 - want3[n:]).max() <= 1e-8 * np.abs(want3).max()
    assert 0.5 * its <= res[0][4]["cg_iters"] / 2 <= 2.5 * its      # same algorithm: comparable iteration counts
    from oracle import pyoracle
    if pyoracle.ref_available():
        ref = pyoracle.load_ref()
        T = ref._scs_types
        prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
        wr = ref.scs_init_lin_sys_work(C.byref(prob.matA), None, dr.ctypes.data_as(T.fp))
        orf = b.copy()
        assert ref.scs_solve_lin_sys(wr, orf.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp), 1e-11) == 0
        ref.scs_free_lin_sys_work(wr)
        assert np.abs(np.concatenate([res[0][0], y]) - orf).max() <= 1e-7 * np.abs(orf).max()


def test_one_rank_over_rccl_c_api_and_the_allreduce_timed_at_the_headline_n(capsys):
    import json
    lib = capi.load("libscsamd.so")
    from scs_amd import problems
    n, m = 1000000, 2000000
    pr = problems.random_socp(n, m, 10, seed=1234)
    dr = probgen.diag_r(n, m, z=0)
    rng = np.random.default_rng(5)
    b = rng.uniform(-1, 1, n + m)
    s = rng.uniform(-1, 1, n) * 0.1
    want, its = _unsplit(pr["A"], dr, b, s, 1e-9)
    S = shard.NativeShardedLinSys(pr["A"], dr, world=1, rank=0, lib=lib)   # ncclCommInitRank with one rank
    S.profiling(True)
    x, y = S.solve(b, s, tol=1e-9)
    st = S.stats()
    S.close()
    sc = np.abs(want).max()
    assert np.abs(x - want[:n]).max() <= 1e-8 * sc and np.abs(y - want[n:]).max() <= 1e-8 * max(sc, np.abs(want[n:]).max())
    assert abs(st["cg_iters"] - its) <= 2                            # world = 1: the same arithmetic up to the grouping of the p.Gp partials
    assert st["allreduces"] >= its and st["allreduces_timed"] > 5 and st["allreduce_mean_us"] > 0
    with capsys.disabled():
        print("\nNATIVE_SHARD " + json.dumps(dict(n=n, world=1, backend="rccl", bytes_per_allreduce=8 * n, **st)))
