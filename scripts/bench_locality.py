#!/usr/bin/env python3
"""SpMV roofline fraction of the headline kernel on matrices WITH column locality: the headline sizes,
cones and data law on a banded pattern (scs_amd/problems.py banded_rows), for several band widths, with the plain and the
software-pipelined instantiation of csr_wave_kernel (SCS_AMD_WR_PIPE=0|1; unset = the library's own choice).
One JSON line per (band, mode).   python scripts/bench_locality.py [--n 1000000] [--bands 4096,65536,0]"""
import argparse, json, os, sys, time
os.environ["SCS_AMD_ALLOW_ENV_HOOKS"] = "1"  # A/B script: the measurement variants of scs_amd/csrc/options.h are set through the environment
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--bands", default="4096,65536,0")
ap.add_argument("--modes", default="auto,0,1")
a = ap.parse_args()
import torch
import bench
args = argparse.Namespace(max_iters=20000)
for band in [int(b) for b in a.bands.split(",")]:
    for mode in a.modes.split(","):
        if mode == "auto":
            os.environ.pop("SCS_AMD_WR_PIPE", None)
        else:
            os.environ["SCS_AMD_WR_PIPE"] = mode
        s = bench.HipSolver(args, 0, 0, a.n, 2 * a.n, 10, 1234, 0, 1e-4, band=band or None)
        s.begin(); s.steps(10)
        st0 = s.stats(); s.profiling(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.steps(30)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        st1 = s.stats(); s.profiling(False); s.end(); s.close()
        nl, ms = st1["spmv_launches"] - st0["spmv_launches"], st1["spmv_ms"] - st0["spmv_ms"]
        cg = st1["cg_iters"] - st0["cg_iters"]
        avg = ms / nl * 1e-3
        bps = st1["spmv_bytes"] / 2.0
        print(json.dumps(dict(band=band, pipe=mode, spmv_avg_us=avg * 1e6, frac_of_8TBs=bps / avg / 1e9 / 8000.0, us_per_cg_iter=1e6 * el / cg if cg else None,
                              cg_its_per_admm_iter=cg / 30.0, launches_timed=int(nl))), flush=True)
