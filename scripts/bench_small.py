"""Launch-bound sizes (BASELINE configs[0] and the sizes between it and configs[3]): ADMM iterations/s and
microseconds per CG iteration of libscsamd.so against the reference CPU library on the same problems.
One JSON line per size.  `--env KEY=VAL` entries are set before the library loads (path switches)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1000,3000,10000,20000,50000,200000")
    ap.add_argument("--max-iters", type=int, default=300)
    ap.add_argument("--cpu", action="store_true", help="also time oracle/_ref/libscsindir_ref.so (1 thread)")
    ap.add_argument("--env", action="append", default=[])
    a = ap.parse_args()
    for kv in a.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    from scs_amd import capi, problems
    lib = capi.load("libscsamd.so")
    ref = None
    if a.cpu:
        from oracle import pyoracle
        ref = pyoracle.load_ref() if pyoracle.ref_available() else None
    for n in [int(s) for s in a.sizes.split(",")]:
        # configs[0] is the reference's own test shape (test/random_socp_prob.c: m = 3n, col_nnz = ceil(sqrt(n)));
        # the larger sizes use the bench recipe (m = 2n, 10 nonzeros per column)
        if n <= 1000:
            data = problems.random_socp(n, m=3 * n, col_nnz=int(n ** 0.5 + 0.999), seed=n)
        else:
            data = problems.random_socp(n, seed=n)
        cone = data["cone"]
        prob = capi.Problem(data["A"], data["b"], data["c"], cone)
        over = dict(max_iters=a.max_iters, eps_abs=1e-9, eps_rel=1e-9, acceleration_lookback=0, verbose=0)
        capi.solve(lib, prob, **over)  # warm-up (graph capture, allocator)
        t0 = time.time()
        out = capi.solve(lib, prob, want_stats=True, **over)
        wall = time.time() - t0
        info, st = out["info"], out["stats"]
        rec = dict(n=n, m=prob.m, nnz=int(prob.sparse().nnz), iters=info["iter"], status=info["status"],
                   solve_ms=round(info["solve_time"], 2), setup_ms=round(info["setup_time"], 2), wall_s=round(wall, 3),
                   admm_it_per_s=round(info["iter"] / (info["solve_time"] / 1e3), 1),
                   cg_its_per_admm=round(st["cg_iters"] / max(info["iter"], 1), 2),
                   us_per_cg_it=round(1e3 * info["lin_sys_time"] / max(st["cg_iters"], 1), 2),
                   lin_sys_ms=round(info["lin_sys_time"], 2), cone_ms=round(info["cone_time"], 2))
        if ref is not None:
            rp = capi.Problem(data["A"], data["b"], data["c"], cone, T=ref._scs_types)
            r = capi.solve(ref, rp, **over)
            rec["cpu_admm_it_per_s"] = round(r["info"]["iter"] / (r["info"]["solve_time"] / 1e3), 1)
            rec["cpu_iters"] = r["info"]["iter"]
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
