#!/usr/bin/env python3
"""PSD projection cost by block order k: the LDS Jacobi path (k <= 92), the
global-memory path (k > 92) and the many-tiny-cones regime, beside the reference's LAPACK dsyevr
(src/cones.c:999-1067) on the host.  For each k a small SDP with B blocks of order k is solved for
a fixed number of iterations by both libraries; the figure is the mean time of ONE projection of all
B blocks (HIP events around the cone kernels / the reference's cone_time).

    python scripts/bench_psd_sizes.py [--ref]
"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scs_amd import capi, problems

ap = argparse.ArgumentParser()
ap.add_argument("--ref", action="store_true")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--check-reps", type=int, default=5)
ap.add_argument("--cases", default="4x2000,8x1000,16x500,32x200,50x200,64x128,92x64,128x32,200x16,256x8,512x4,1024x1")
a = ap.parse_args()
amd = capi.load("libscsamd.so")
ref = None
if a.ref:
    from oracle import pyoracle
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    ref = pyoracle.load_ref()
rows = []
for case in a.cases.split(","):
    k, B = (int(v) for v in case.split("x"))
    pr = problems.random_sdp(200, B, k, 2, 4, seed=1)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    kw = dict(verbose=0, acceleration_lookback=0, max_iters=a.iters, eps_abs=1e-12, eps_rel=1e-12)
    t0 = time.time()
    r = capi.solve(amd, prob, want_stats=True, profiling=True, **kw)
    st = r["stats"]
    row = dict(k=k, blocks=B, gpu_ms_per_projection=st["cone_ms"] / max(st["cone_projs"], 1), gpu_us_per_block=1e3 * st["cone_ms"] / max(st["cone_projs"], 1) / B,
               flop_model=10.0 * k ** 3 * B, psd_unconverged=st.get("psd_unconverged"), wall_s=time.time() - t0)
    row["gpu_gflops_model"] = row["flop_model"] / (row["gpu_ms_per_projection"] * 1e-3) / 1e9
    # round 6 ("times 64x128 without checking the result"): the same B blocks of order k through the same kernels
    # (scs_amd_cone_proj_dual: cold, then a drifting warm-started sequence like consecutive ADMM iterates) against numpy's eigh
    cone = dict(s=[k] * B)
    mc = capi.cone_rows(cone)
    kc = capi.make_cone(cone)
    wc = amd.scs_amd_cone_init(C.byref(kc), mc, None)
    rng = np.random.default_rng(k)
    v = rng.standard_normal(mc)
    worst = 0.0
    for rep in range(a.check_reps):
        v = v + (0.3 if rep < 2 else 1e-3) * rng.standard_normal(mc)
        got = v.copy()
        assert amd.scs_amd_cone_proj_dual(wc, got.ctypes.data_as(capi.T64.fp), None) == 0
        want = problems.proj_dual_cone_np(v, cone)
        worst = max(worst, float(np.abs(got - want).max() / max(1.0, np.abs(want).max())))
    amd.scs_amd_cone_finish(wc)
    row["max_err_vs_numpy_eigh"] = worst
    row["checked_projections"] = a.check_reps
    if worst > 1e-10:
        row["CHECK_FAILED"] = True
    if ref is not None and k * k * B <= 1100000:
        rr = capi.solve(ref, prob, **kw)["info"]
        row["cpu_ms_per_projection"] = rr["cone_time"] / max(rr["iter"], 1)
        row["speedup"] = row["cpu_ms_per_projection"] / row["gpu_ms_per_projection"]
    rows.append(row)
    print(json.dumps(row), flush=True)
