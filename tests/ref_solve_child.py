"""Child process of the full-size trajectory tests: the reference (one of oracle/_ref's flavours) on a generated headline-family
problem for a capped number of ADMM iterations, in a process of its own so that OMP_NUM_THREADS / OMP_WAIT_POLICY are read by a
fresh libgomp.  Writes info + (x, y, s) to an .npz.  Test infrastructure only (CPU).

    python tests/ref_solve_child.py flavour n m col_nnz seed max_iters out.npz
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if __name__ == "__main__":
    flavour = sys.argv[1]
    n, m, col_nnz, seed, max_iters = (int(v) for v in sys.argv[2:7])
    out = sys.argv[7]
    from oracle import pyoracle
    from scs_amd import capi, problems
    ref = pyoracle.load_ref(flavour)
    pr = problems.random_socp(n, m, col_nnz, seed=seed)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    t0 = time.time()
    r = capi.solve(ref, prob, verbose=0, acceleration_lookback=0, max_iters=max_iters)
    print("reference %s: %d iterations in %.1f s" % (flavour, r["info"]["iter"], time.time() - t0), flush=True)
    np.savez(out, x=r["x"], y=r["y"], s=r["s"], info=json.dumps(r["info"]))
