"""One small invocation of the hot path on cuda:0, checked against the oracle
(the plain-C restatement; only smoke/tests/bench's cpu_baseline may touch oracle/)."""
import numpy as np


def run():
    from scs_amd import capi, problems
    lib = capi.load("libscsamd.so")  # raises if the HIP extension is missing: no CPU fallback
    if lib.scs_amd_device_count() <= 0:
        raise RuntimeError("smoke: no HIP device visible")
    assert lib.scs_amd_set_device(0) == 0
    cone = dict(z=20, l=60, bl=[-1.0] * 9, bu=[1.0] * 9, q=[5, 40, 3], s=[6, 4])
    m = capi.cone_rows(cone)
    pr = problems.random_cone_prob(60, m, 6, cone, seed=7)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    ra = capi.solve(lib, prob, verbose=0, acceleration_lookback=0, cg_tol_override=1e-12, want_stats=True)
    from oracle import pyoracle
    ro = pyoracle.oracle_solve(prob, cg_tol_override=1e-12)
    ia, io = ra["info"], ro["info"]
    assert ia["status_val"] == io["status_val"] == 1, (ia, io)
    assert ia["iter"] == io["iter"], (ia["iter"], io["iter"])
    for k in ("pobj", "dobj", "res_pri", "res_dual", "gap"):
        assert abs(ia[k] - io[k]) <= 1e-6 * max(abs(io[k]), 1e-3), (k, ia[k], io[k])
    assert np.abs(ra["x"] - ro["x"]).max() <= 1e-6 * max(1.0, np.abs(ro["x"]).max())
    print(f"smoke ok: {ia['iter']} ADMM iterations, {ra['stats']['cg_iters']} CG iterations, "
          f"pobj {ia['pobj']:.9f} (oracle {io['pobj']:.9f}), solver {ia['lin_sys_solver']}")


if __name__ == "__main__":
    run()
