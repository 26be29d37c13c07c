"""Solve one random SOCP with the HIP library and (optionally) the reference; print both infos."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scs_amd import capi, problems

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--m", type=int, default=0)
ap.add_argument("--col-nnz", type=int, default=32)
ap.add_argument("--seed", type=int, default=1234)
ap.add_argument("--ref", action="store_true")
ap.add_argument("--aa", type=int, default=0)
ap.add_argument("--verbose", type=int, default=0)
ap.add_argument("--max-iters", type=int, default=100000)
ap.add_argument("--q-fixed", type=int, default=0)
a = ap.parse_args()
m = a.m or 3 * a.n
t0 = time.time()
pr = problems.random_socp(a.n, m, a.col_nnz, seed=a.seed, q_fixed=a.q_fixed or None)
prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
print(f"gen {time.time()-t0:.1f}s cones: z={pr['cone']['z']} l={pr['cone']['l']} nsoc={len(pr['cone']['q'])}", flush=True)
kw = dict(verbose=a.verbose, acceleration_lookback=a.aa, max_iters=a.max_iters)
amd = capi.load("libscsamd.so")
t0 = time.time()
ra = capi.solve(amd, prob, want_stats=True, **kw)
print("amd", json.dumps(ra["info"]), json.dumps(ra.get("stats")), f"wall {time.time()-t0:.2f}s", flush=True)
if a.ref:
    from oracle import pyoracle
    ref = pyoracle.load_ref()
    t0 = time.time()
    rr = capi.solve(ref, prob, **kw)
    print("ref", json.dumps(rr["info"]), f"wall {time.time()-t0:.2f}s", flush=True)
    import numpy as np
    print("max|x_amd - x_ref| =", np.abs(ra["x"] - rr["x"]).max(), " |x|max", np.abs(rr["x"]).max())
