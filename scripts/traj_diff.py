"""Find the first ADMM iteration at which the HIP path and the reference diverge:
run both with max_iters = K for increasing K and compare the returned iterates."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scs_amd import capi, problems
from oracle import pyoracle

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--col-nnz", type=int, default=32)
ap.add_argument("--normalize", type=int, default=1)
ap.add_argument("--adaptive", type=int, default=1)
ap.add_argument("--ks", default="1,2,3,5,10,25,26,27,50,51,52,100,101,102,126,127,151,200,300")
a = ap.parse_args()
pr = problems.random_socp(a.n, 3 * a.n, a.col_nnz, seed=1234)
prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
amd, ref = capi.load("libscsamd.so"), pyoracle.load_ref()
for K in [int(k) for k in a.ks.split(",")]:
    kw = dict(verbose=0, acceleration_lookback=0, max_iters=K, normalize=a.normalize, adaptive_scale=a.adaptive,
              eps_abs=1e-12, eps_rel=1e-12)
    ra, rr = capi.solve(amd, prob, **kw), capi.solve(ref, prob, **kw)
    dx = np.abs(ra["x"] - rr["x"]).max() / max(1e-300, np.abs(rr["x"]).max())
    dy = np.abs(ra["y"] - rr["y"]).max() / max(1e-300, np.abs(rr["y"]).max())
    ds = np.abs(ra["s"] - rr["s"]).max() / max(1e-300, np.abs(rr["s"]).max())
    ia, ir = ra["info"], rr["info"]
    print(f"K={K:4d} dx={dx:.2e} dy={dy:.2e} ds={ds:.2e} pobj {ia['pobj']:.10e} {ir['pobj']:.10e} "
          f"rp {ia['res_pri']:.6e} {ir['res_pri']:.6e} scale {ia['scale']:.6g} {ir['scale']:.6g} "
          f"upd {ia['scale_updates']} {ir['scale_updates']}", flush=True)
