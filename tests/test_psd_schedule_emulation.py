"""A numpy restatement of the SCHEDULE of the blocked Jacobi iteration for PSD blocks beyond the LDS path
(scs_amd/csrc/psd_big.h: `bj_pair`, `bj_pair_sched`, `bj_inner_pair`, `k_bj_inner`, `k_bj_update`) -- test infrastructure, CPU
only: tile-level, not lane-level.  It pins what the kernels' orchestration relies on:

 * the tournament over block columns pairs every two block columns exactly once per sweep and covers all columns in every step;
 * the every-pair-once schedule (a within-block pass, then cross pairs only) rotates every index pair exactly once per sweep;
 * with the update as the kernels do it (diagonal tile = the inner sweep's S', other tiles Q_P' A[P,Q] Q_Q, V[:,Q] Q_Q, untouched
   tiles copied, padding rows / columns never rotated) the iteration converges to the eigen-decomposition: reconstruction and
   eigenvalues to 1e-11, orthogonality to 1e-12, in about ten sweeps -- the numbers DESIGN.md quotes for this schedule.
The GPU kernels themselves are compared with LAPACK through the reference in tests/test_cones_shim_gpu.py."""
import itertools

import numpy as np
import pytest

B, W = 32, 64


def bj_pair(i, step, nbc):
    I = 0 if i == 0 else 1 + ((i - 1 + step) % (nbc - 1))
    J = 1 + ((nbc - 2 - i + step) % (nbc - 1))
    return (I, J) if I < J else (J, I)


def bj_pair_sched(i, ostep, nbc, cross):
    if not cross:
        return bj_pair(i, ostep, nbc)
    return (2 * i, 2 * i + 1) if ostep == 0 else bj_pair(i, ostep - 1, nbc)


def bj_inner_pair(i, st, kind):
    if kind == 2:
        return i, B + ((i + st) & (B - 1))
    n, j, off = (B, i & (B // 2 - 1), B if i >= B // 2 else 0) if kind == 1 else (W, i, 0)
    p = 0 if j == 0 else 1 + ((j - 1 + st) % (n - 1))
    q = 1 + ((n - 2 - j + st) % (n - 1))
    p, q = (p, q) if p < q else (q, p)
    return p + off, q + off


def gidx(IJ):
    return np.r_[IJ[0] * B:(IJ[0] + 1) * B, IJ[1] * B:(IJ[1] + 1) * B]


@pytest.mark.parametrize("nbc", [2, 4, 6, 8, 32])
def test_block_column_tournament_meets_every_pair_once_and_covers_all_columns(nbc):
    met = set()
    for step in range(nbc - 1):
        pairs = [bj_pair(i, step, nbc) for i in range(nbc // 2)]
        assert sorted(itertools.chain.from_iterable(pairs)) == list(range(nbc))
        for pr in pairs:
            assert pr not in met
            met.add(pr)
    assert len(met) == nbc * (nbc - 1) // 2


def test_every_pair_once_schedule_rotates_every_index_pair_exactly_once_per_sweep():
    nbc = 6
    seen = {}
    for ostep in range(nbc):  # the within pass, then the nbc - 1 tournament steps
        kind = 1 if ostep == 0 else 2
        for pi in range(nbc // 2):
            g = gidx(bj_pair_sched(pi, ostep, nbc, True))
            for st in range(B - 1 if kind == 1 else B):
                pairs = [bj_inner_pair(i, st, kind) for i in range(B)]
                assert sorted(itertools.chain.from_iterable(pairs)) == list(range(W))  # 32 disjoint pairs per inner step
                for p, q in pairs:
                    key = (int(g[p]), int(g[q]))
                    assert key[0] < key[1]
                    seen[key] = seen.get(key, 0) + 1
    K = nbc * B
    assert len(seen) == K * (K - 1) // 2 and set(seen.values()) == {1}


def _inner(S, g, k, thr, kind):
    S, Q, offmax, rotated = S.copy(), np.eye(W), 0.0, False
    for st in range({0: W - 1, 1: B - 1, 2: B}[kind]):
        J, rot = np.eye(W), []
        for i in range(B):
            p, q = bj_inner_pair(i, st, kind)
            apq = S[p, q]
            if g[q] < k:
                offmax = max(offmax, abs(apq))
                if abs(apq) > thr:
                    d, b = S[q, q] - S[p, p], 2 * apq
                    qq = 1 / np.sqrt(d * d + b * b)          # the two-rsqrt form of cones.hip `jacobi_cs`
                    u = 0.5 + 0.5 * abs(d) * qq
                    rc = 1 / np.sqrt(u)
                    c, s = u * rc, (b if d >= 0 else -b) * 0.5 * qq * rc
                    J[p, p] = J[q, q] = c
                    J[p, q], J[q, p] = s, -s
                    rot.append((p, q))
        if rot:
            rotated = True
            S = J.T @ S @ J
            for p, q in rot:
                S[p, q] = S[q, p] = 0.0   # a rotated pair's own entry: exact zero
            Q = Q @ J
    return Q, S, rotated, offmax


def _project(Ain, k, cross, scan=False):
    K64 = (k + W - 1) // W * W
    nbc, npairs = K64 // B, K64 // W
    A = [np.zeros((K64, K64)), np.zeros((K64, K64))]
    A[0][:k, :k] = Ain
    V, cur, thr = np.eye(K64), 0, 1e-15 * np.linalg.norm(Ain) / k
    for sweep in range(1, 31):
        off = 0.0
        for ostep in range(nbc if cross else nbc - 1):
            kind = 0 if not cross else (1 if ostep == 0 else 2)
            old, new = A[cur], A[cur ^ 1]
            pairs = [gidx(bj_pair_sched(pi, ostep, nbc, cross)) for pi in range(npairs)]
            res = [_inner(old[np.ix_(g, g)], g, k, thr, kind) for g in pairs]
            off = max([off] + [r[3] for r in res])
            for P in range(npairs):
                for Qp in range(npairs):
                    gp, gq, fP, fQ = pairs[P], pairs[Qp], res[P][2], res[Qp][2]
                    if P == Qp and fP:
                        new[np.ix_(gp, gq)] = res[P][1]
                    elif not fP and not fQ:
                        new[np.ix_(gp, gq)] = old[np.ix_(gp, gq)]
                    else:
                        QP = res[P][0] if fP else np.eye(W)
                        QQ = res[Qp][0] if fQ else np.eye(W)
                        new[np.ix_(gp, gq)] = QP.T @ (old[np.ix_(gp, gq)] @ QQ)
            for Qp in range(npairs):
                if res[Qp][2]:
                    V[:, pairs[Qp]] = V[:, pairs[Qp]] @ res[Qp][0]
            cur ^= 1
        if off <= thr:
            return A[cur], V, sweep
        if scan:  # round 6 (psd_big.h, k_bp_offscan / the last update's left_bits): would the NEXT sweep rotate anything?
            left = np.abs(A[cur][:k, :k] - np.diag(np.diag(A[cur][:k, :k]))).max()
            if left <= thr:
                return A[cur], V, sweep
    return A[cur], V, 31


@pytest.mark.parametrize("k,cross", [(100, True), (100, False), (131, True)])
def test_emulated_blocked_iteration_converges_to_the_eigen_decomposition(k, cross):
    rng = np.random.default_rng(k)
    M = rng.standard_normal((k, k))
    Ain = (M + M.T) / 2
    D, V, sweeps = _project(Ain, k, cross)
    lam, Vk = np.diag(D)[:k], V[:k, :k]
    assert sweeps <= 13
    assert np.abs((Vk * lam) @ Vk.T - Ain).max() <= 1e-11
    assert np.abs(np.sort(lam) - np.linalg.eigvalsh(Ain)).max() <= 1e-11
    assert np.abs(Vk.T @ Vk - np.eye(k)).max() <= 1e-12
    if V.shape[0] > k:  # padding rows / columns were never rotated
        assert np.abs(V[k:, :k]).max() == 0.0 and np.abs(V[:k, k:]).max() == 0.0


def _fused_subproblem(old, pairs_k, res_k, g_next):
    """The 64 x 64 subproblem of the NEXT step's pair (I', J') as k_bj_fused's inner workgroups form it from what the update of step k
    reads -- the matrix before the step (`old`), and Q / S' / flag of step k's pairs (`res_k`): quadrants [I', I'] and [J', J'] are pieces
    of the diagonal tiles of step k (S' of the pair that held the block column, or A itself if that pair did not rotate), [I', J'] is a
    32 x 32 piece of Q_Pa' A[Pa, Pb] Q_Pb with the update's orientation Pa <= Pb (transposed when I' sat in the larger pair)."""
    I, J = int(g_next[0]) // B, int(g_next[B]) // B

    def where(X):
        for pi, g in enumerate(pairs_k):
            if g[0] // B == X:
                return pi, 0
            if g[B] // B == X:
                return pi, 1
        raise AssertionError("block column not in any pair")

    def diag_tile(P):
        Qm, Sp, rot, _ = res_k[P]
        return Sp if rot else old[np.ix_(pairs_k[P], pairs_k[P])]

    (PI, hI), (PJ, hJ) = where(I), where(J)
    S = np.zeros((W, W))
    sl = lambda h: slice(h * B, (h + 1) * B)
    S[:B, :B] = diag_tile(PI)[sl(hI), sl(hI)]
    S[B:, B:] = diag_tile(PJ)[sl(hJ), sl(hJ)]
    if PI == PJ:
        X = diag_tile(PI)[sl(hI), sl(hJ)]
    else:
        swap = PI > PJ
        Pa, Pb, ha, hb = (PJ, PI, hJ, hI) if swap else (PI, PJ, hI, hJ)
        Qa = res_k[Pa][0] if res_k[Pa][2] else np.eye(W)
        Qb = res_k[Pb][0] if res_k[Pb][2] else np.eye(W)
        O = Qa[:, sl(ha)].T @ (old[np.ix_(pairs_k[Pa], pairs_k[Pb])] @ Qb[:, sl(hb)])
        X = O.T if swap else O
    S[:B, B:] = X
    S[B:, :B] = X.T
    return S


@pytest.mark.parametrize("k,cross", [(200, True), (131, False)])
def test_fused_step_forms_the_subproblem_the_update_writes(k, cross):
    """The index bookkeeping of k_bj_fused (psd_big.h): for every outer step of two sweeps, the subproblem assembled from pre-update
    data equals the diagonal block of the updated matrix for the next step's pairing."""
    rng = np.random.default_rng(k)
    M = rng.standard_normal((k, k))
    Ain = (M + M.T) / 2
    K64 = (k + W - 1) // W * W
    nbc, npairs = K64 // B, K64 // W
    A = [np.zeros((K64, K64)), np.zeros((K64, K64))]
    A[0][:k, :k] = Ain
    cur, thr = 0, 1e-15 * np.linalg.norm(Ain) / k
    osteps = nbc if cross else nbc - 1
    checked = 0
    for sweep in range(2):
        for ostep in range(osteps):
            kind = 0 if not cross else (1 if ostep == 0 else 2)
            old, new = A[cur], A[cur ^ 1]
            pairs = [gidx(bj_pair_sched(pi, ostep, nbc, cross)) for pi in range(npairs)]
            res = [_inner(old[np.ix_(g, g)], g, k, thr, kind) for g in pairs]
            for P in range(npairs):
                for Qp in range(npairs):
                    gp, gq, fP, fQ = pairs[P], pairs[Qp], res[P][2], res[Qp][2]
                    if P == Qp and fP:
                        new[np.ix_(gp, gq)] = res[P][1]
                    elif not fP and not fQ:
                        new[np.ix_(gp, gq)] = old[np.ix_(gp, gq)]
                    else:
                        QP = res[P][0] if fP else np.eye(W)
                        QQ = res[Qp][0] if fQ else np.eye(W)
                        new[np.ix_(gp, gq)] = QP.T @ (old[np.ix_(gp, gq)] @ QQ)
            nstep = (ostep + 1) % osteps  # the last launch of a sweep runs the first inner sweep of the next one
            for pi in range(npairs):
                g_next = gidx(bj_pair_sched(pi, nstep, nbc, cross))
                got = _fused_subproblem(old, pairs, res, g_next)
                want = new[np.ix_(g_next, g_next)]
                assert np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max()), (sweep, ostep, pi)
                checked += 1
            cur ^= 1
    assert checked == 2 * osteps * npairs


def test_cross_sweep_items_cover_the_upper_triangle_once_per_step_and_the_look_ahead_reads_the_right_blocks():
    """The lane bookkeeping of bj_inner_sweep_cross (psd_big.h), restated: in every step the 496 off-diagonal blocks (P < Q) and the 32
    diagonal blocks write every entry of the UPPER triangle (r <= c in local order) exactly once -- rows {P, q(P)} x columns {Q, q(Q)},
    q(i) = 32 + (i + st) mod 32, with S[q(P)][Q] kept at its mirror (Q, q(P)) and S[q(P)][q(Q)] at (min, max) -- and the ten entries a
    look-ahead lane reads are the operands of the three blocks whose n11 / n22 / n12-or-n21 are S'[p'][p'], S'[q'][q'], S'[p'][q'] of
    ITS next pair (p' = i, q' = 32 + (i + st + 1) mod 32)."""
    q = lambda i, st: B + ((i + st) % B)
    for st in range(B):
        seen = np.zeros((W, W), dtype=int)
        for Q in range(B):
            for P in range(Q + 1):
                qP, qQ = q(P, st), q(Q, st)
                lo, hi = min(qP, qQ), max(qP, qQ)
                offs = [(P, Q), (P, qQ), (Q, qP), (lo, hi)] if P < Q else [(P, P), (P, qP), (qP, qP)]
                for r, c in offs:
                    assert r <= c
                    seen[r, c] += 1
        assert (seen[np.triu_indices(W)] == 1).all() and seen[np.tril_indices(W, -1)].sum() == 0
        # the full-matrix meaning of a block's four operands: a11 = S[P][Q], a12 = S[P][qQ], a21 = S[qP][Q], a22 = S[qP][qQ]
        for i in range(B):
            j = (i + 1) % B
            qa, qn = q(i, st), q(i, st + 1)
            assert qn == q(j, st)  # next step's partner of i is this step's partner of pair i + 1
            fP, fQ = min(i, j), max(i, j)
            qPf, qQf = (qa, qn) if i < j else (qn, qa)
            ao = [(i, i), (i, qa), (qa, qa), (j, j), (j, qn), (qn, qn), (fP, fQ), (fP, qQf), (fQ, qPf), (min(qPf, qQf), max(qPf, qQf))]
            assert all(r <= c for r, c in ao)
            # block (i, i): rows {i, qa} x cols {i, qa}: n11 is entry (i, i) = (p', p')
            # block (j, j): rows {j, qn} x cols {j, qn}: n22 is entry (qn, qn) = (q', q')
            # block (fP, fQ): rows {fP, q(fP)} x cols {fQ, q(fQ)}: n12 = (fP, q(fQ)), n21 = (q(fP), fQ)
            want = (i, qn)  # (p', q')
            got = (fP, qQf) if i < j else (fQ, qPf)  # n12's row/col if i < j, else the mirror of n21 = (q(fP), fQ) -> (fQ, q(fP))
            assert got == want, (st, i, got, want)


@pytest.mark.parametrize("K2", [2, 4, 6, 50, 64, 92])
def test_incremental_round_robin_positions_equal_the_closed_form(K2):
    """k_psd_jacobi (cones.hip) advances the two positions of a lane's pair by one per step (wrapping from K2 - 1 to 1, position 0
    fixed) instead of evaluating pair i of step s = (0 or 1 + (i - 1 + s) mod (K2 - 1), 1 + (K2 - 2 - i + s) mod (K2 - 1)): the same
    pairs, every index pair exactly once per sweep."""
    met = set()
    for i in range(K2 // 2):
        a, b = i, K2 - 1 - i
        for step in range(K2 - 1):
            p = 0 if i == 0 else 1 + ((i - 1 + step) % (K2 - 1))
            q = 1 + ((K2 - 2 - i + step) % (K2 - 1))
            assert (a, b) == (p, q)
            met.add((min(p, q), max(p, q)))
            if i != 0:
                a = 1 if a == K2 - 1 else a + 1
            b = 1 if b == K2 - 1 else b + 1
    assert len(met) == K2 * (K2 - 1) // 2


@pytest.mark.parametrize("k,cross", [(100, True), (70, False)])
def test_closing_by_a_scan_ends_one_sweep_earlier_with_the_same_matrix(k, cross):
    """Round 6: the iteration used to end with a sweep that meets no entry above the threshold, i.e. one that rotates nothing.  Reading
    the largest off-diagonal entry off the matrix after every sweep (psd_big.h: k_bp_offscan, or the last update of the sweep itself)
    ends it exactly one sweep earlier and leaves A and V bit for bit as the closing sweep would have (that sweep changes nothing)."""
    rng = np.random.default_rng(7 * k)
    M = rng.standard_normal((k, k))
    Ain = (M + M.T) / 2
    D0, V0, s0 = _project(Ain, k, cross)
    D1, V1, s1 = _project(Ain, k, cross, scan=True)
    assert s1 == s0 - 1
    assert np.array_equal(D0, D1) and np.array_equal(V0, V1)

