"""Host-side acceptance checks of a returned solution, in the manner of the reference's
`verify_solution_correct` (test/problem_utils.h:107-249): everything is recomputed from the
problem data and the returned (x, y, s) with numpy/scipy -- nothing is taken from the solver
except the `ScsInfo` numbers being checked.  Used by bench.py (`batch.parity`) and the tests;
not part of the solve path.

Checks for status "solved" (problem_utils.h:197-219):
  |res_pri - info.res_pri| < 1e-10, |res_dual - info.res_dual| < 1e-10,
  |gap - info.gap| < 1e-7 (1 + |gap|), |pobj - info.pobj| < 1e-9 (1 + |pobj|), same for dobj,
  |s'y| < 5e-8 max(|s|_inf, |y|_inf), dist(s, K) < 1e-5, dist(y, K*) < 1e-5,
  res_pri < eps_abs + eps_rel max(|b|, |s|, |Ax|)_inf, res_dual < eps_abs + eps_rel max(|c|, |A'y|)_inf,
  gap < eps_abs + eps_rel max(|c'x|, |b'y|).
The info-identity tolerances (first two lines) are the reference's fp64 calibration; `info_tol_scale`
loosens them for fp32 builds.
"""
import numpy as np

from . import problems


def verify_solved(A, b, c, cone, x, y, s, info, eps_abs=1e-4, eps_rel=1e-4, info_tol_scale=1.0):
    """Returns dict(ok=bool, failed=[names], values={...}).  A: scipy sparse (m x n); cone: dict as in capi.make_cone."""
    x, y, s = (np.asarray(v, dtype=np.float64) for v in (x, y, s))
    b, c = np.asarray(b, dtype=np.float64), np.asarray(c, dtype=np.float64)
    A = A.astype(np.float64) if A.dtype != np.float64 else A
    ax, aty = A @ x, A.T @ y
    res_pri = float(np.abs(ax + s - b).max())
    res_dual = float(np.abs(aty + c).max())
    ctx, bty, sty = float(c @ x), float(b @ y), float(s @ y)
    gap = abs(ctx + bty)
    pobj, dobj = ctx, -bty
    # ||s - Pi_K(s)|| = ||Pi_K*(-s)||, ||y - Pi_K*(y)|| = ||Pi_K(-y)|| (problem_utils.h:83-104); Pi_K(v) = v + Pi_K*(-v)
    sdist = float(np.abs(problems.proj_dual_cone_np(-s, cone)).max())
    ydist = float(np.abs(-y + problems.proj_dual_cone_np(y, cone)).max())
    prl = max(np.abs(b).max(), np.abs(s).max(), np.abs(ax).max())
    drl = max(np.abs(c).max(), np.abs(aty).max())
    grl = max(abs(ctx), abs(bty))
    t = info_tol_scale
    checks = {
        "res_pri_identity": abs(res_pri - info["res_pri"]) < 1e-10 * t,
        "res_dual_identity": abs(res_dual - info["res_dual"]) < 1e-10 * t,
        "gap_identity": abs(gap - info["gap"]) < 1e-7 * t * (1 + abs(gap)),
        "pobj_identity": abs(pobj - info["pobj"]) < 1e-9 * t * (1 + abs(pobj)),
        "dobj_identity": abs(dobj - info["dobj"]) < 1e-9 * t * (1 + abs(dobj)),
        "complementary_slackness": abs(sty) < 5e-8 * t * max(np.abs(s).max(), np.abs(y).max()),
        "s_in_K": sdist < 1e-5,
        "y_in_Kstar": ydist < 1e-5,
        "primal_feasible": res_pri < eps_abs + eps_rel * prl,
        "dual_feasible": res_dual < eps_abs + eps_rel * drl,
        "gap_small": gap < eps_abs + eps_rel * grl,
    }
    values = dict(res_pri=res_pri, res_dual=res_dual, gap=gap, pobj=pobj, dobj=dobj, sty=sty, sdist=sdist, ydist=ydist,
                  limit_pri=float(eps_abs + eps_rel * prl), limit_dual=float(eps_abs + eps_rel * drl),
                  limit_gap=float(eps_abs + eps_rel * grl))
    failed = [k for k, v in checks.items() if not v]
    return dict(ok=not failed, failed=failed, values=values)
