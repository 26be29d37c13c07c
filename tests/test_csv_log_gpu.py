"""`log_csv_filename` (reference src/rw.c:686-863): same header, same number of rows and the
same numbers per iteration as the reference build writes, exact CG on both sides so the
trajectories are comparable."""
import os

import numpy as np
import pytest

from oracle import pyoracle
from scs_amd import capi, problems

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")]


def _read(path):
    lines = open(path).read().splitlines()
    head = lines[0]
    rows = []
    for ln in lines[1:]:
        assert ln.endswith(",")
        rows.append([float(t) for t in ln[:-1].split(",")])
    return head, np.array(rows)


@pytest.mark.parametrize("aa", [0, 10])
def test_csv_log_matches_reference(tmp_path, aa):
    ref = pyoracle.load_ref("libscsindir_ref_exactcg.so")
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(150, 450, 8, seed=3)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    fa, fr = str(tmp_path / "amd.csv"), str(tmp_path / "ref.csv")
    kw = dict(verbose=0, acceleration_lookback=aa, max_iters=60, eps_abs=1e-12, eps_rel=1e-12)
    capi.solve(amd, prob, cg_tol_override=1e-12, log_csv_filename=fa.encode(), **kw)
    capi.solve(ref, prob, log_csv_filename=fr.encode(), **kw)
    ha, ra = _read(fa)
    hr, rr = _read(fr)
    assert ha == hr                      # column names, order, trailing comma
    assert ra.shape == rr.shape          # one row per iteration + the final one
    names = ha.rstrip(",").split(",")
    assert np.array_equal(ra[:, 0], rr[:, 0])
    t = names.index("time")
    for j, name in enumerate(names[: ra.shape[1]]):
        if j in (0, t):
            continue
        a, r = ra[:, j], rr[:, j]
        assert np.array_equal(np.isnan(a), np.isnan(r)), name
        ok = ~np.isnan(r)
        scale = np.maximum(1e-9, np.abs(r[ok]))
        # AA's least-squares solve amplifies rounding differences late in the run
        tol = 1e-6 if aa == 0 else 1e-3
        assert np.all(np.abs(a[ok] - r[ok]) <= tol * scale + 1e-12), (name, np.abs(a[ok] - r[ok]).max())
    assert np.all(np.diff(ra[:, t]) >= 0)


def test_csv_log_is_rewritten_by_each_solve(tmp_path):
    amd = capi.load("libscsamd.so")
    pr = problems.random_socp(60, 180, 6, seed=1)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    f = str(tmp_path / "log.csv")
    r = capi.solve(amd, prob, verbose=0, log_csv_filename=f.encode())
    head, rows = _read(f)
    assert rows.shape[0] == r["info"]["iter"] + 1
    assert rows[-1, 0] == r["info"]["iter"]
    np.testing.assert_allclose(rows[-1, 1], r["info"]["res_pri"], rtol=1e-12)
    np.testing.assert_allclose(rows[-1, 23], r["info"]["pobj"], rtol=1e-12)
