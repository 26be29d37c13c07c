"""BASELINE configs[2]: random SDP, 200 PSD blocks of 50x50 + box cone, on one GPU.
Prints ScsInfo of the HIP solve, the per-projection cone time (HIP events), and -- with
--ref -- the reference CPU solve of the same problem for comparison."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scs_amd import capi, problems

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--blocks", type=int, default=200)
ap.add_argument("--k", type=int, default=50)
ap.add_argument("--bsize", type=int, default=1001)
ap.add_argument("--aa", type=int, default=0)
ap.add_argument("--ref", action="store_true")
ap.add_argument("--exact", action="store_true")
a = ap.parse_args()
t0 = time.time()
pr = problems.random_sdp(a.n, a.blocks, a.k, a.bsize, 10, seed=1234)
prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
print(f"gen {time.time()-t0:.1f}s n={prob.n} m={prob.m} nnz={len(prob.Ax)}", flush=True)
amd = capi.load("libscsamd.so")
kw = dict(verbose=0, acceleration_lookback=a.aa)
t0 = time.time()
ra = capi.solve(amd, prob, want_stats=True, cg_tol_override=1e-12 if a.exact else None, **kw)
st = ra["stats"]
print("amd", json.dumps(ra["info"]), f"wall {time.time()-t0:.2f}s")
print(f"cone: {st['cone_ms']/max(st['cone_projs'],1):.3f} ms per projection over {st['cone_projs']} timed projections "
      f"({a.blocks} blocks of {a.k}x{a.k}); cg its {st['cg_iters']}", flush=True)
if a.ref:
    from oracle import pyoracle
    ref = pyoracle.load_ref("libscsindir_ref_exactcg.so" if a.exact else "libscsindir_ref.so")
    t0 = time.time()
    rr = capi.solve(ref, prob, **kw)
    print("ref", json.dumps(rr["info"]), f"wall {time.time()-t0:.2f}s")
    print(f"ref cone: {rr['info']['cone_time']/max(rr['info']['iter'],1):.3f} ms per projection")
