#!/usr/bin/env python3
"""SpMV time per product and wall time per CG iteration of one headline-family problem under several instantiations of the wave
kernel (environment switches read by scs_init), for the decision which one the library picks by itself.
    python scripts/bench_spmv_modes.py --cases 1000000:f64:0,200000:f64:0,4000000:f32:0,1000000:f64:1024 --modes plain,ls16,ls8
One JSON line per (case, mode)."""
import argparse, json, os, sys, time
os.environ["SCS_AMD_ALLOW_ENV_HOOKS"] = "1"  # A/B script: the measurement variants of scs_amd/csrc/options.h are set through the environment
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="1000000:f64:0")
ap.add_argument("--modes", default="plain,ls16")
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
import torch
import bench
MODES = {
    "auto": {},
    "plain": {"SCS_AMD_WR_LOCKSTEP": "0"},
    "ls16": {"SCS_AMD_WR_LOCKSTEP": "1", "SCS_AMD_WR_LS_WPB": "16", "SCS_AMD_WR_WPC": "16", "SCS_AMD_WR_LS_BARRIERS": "4"},
    "ls16b1": {"SCS_AMD_WR_LOCKSTEP": "1", "SCS_AMD_WR_LS_WPB": "16", "SCS_AMD_WR_WPC": "16", "SCS_AMD_WR_LS_BARRIERS": "1"},
    "ls8": {"SCS_AMD_WR_LOCKSTEP": "1", "SCS_AMD_WR_LS_WPB": "8", "SCS_AMD_WR_LS_BARRIERS": "4"},
}
TOKENS = {"wpc": "SCS_AMD_WR_WPC", "nnz": "SCS_AMD_WR_NNZ", "pipe": "SCS_AMD_WR_PIPE", "ls": "SCS_AMD_WR_LOCKSTEP", "wave": "SCS_AMD_WAVEROWS",
          "bars": "SCS_AMD_WR_LS_BARRIERS", "wpb": "SCS_AMD_WR_LS_WPB", "ro": "SCS_AMD_REORDER", "rh": "SCS_AMD_REORDER_HOME", "rs": "SCS_AMD_REORDER_STRIDE", "rb": "SCS_AMD_REORDER_BLOCK", "cgthree": "SCS_AMD_CG3", "graph": "SCS_AMD_GRAPH", "lso": "SCS_AMD_WR_LS_ORDER", "nt": "SCS_AMD_VEC_NT", "dir": "SCS_AMD_DIR_MODE"}


def mode_env(mode):
    """a named mode, or tokens joined by '+': plain+wpc16+nnz512+pipe1 (token = key followed by its value)"""
    if mode in MODES:
        return dict(MODES[mode])
    env = {}
    for tok in mode.split("+"):
        if tok in MODES:
            env.update(MODES[tok])
            continue
        for k, var in TOKENS.items():
            if tok.startswith(k) and tok[len(k):].isdigit():
                env[var] = tok[len(k):]
                break
        else:
            raise SystemExit(f"unknown mode token {tok!r}")
    return env


KEYS = sorted({k for m in MODES.values() for k in m} | set(TOKENS.values()))
args = argparse.Namespace(max_iters=20000)
for case in a.cases.split(","):
    n, dtype, band = case.split(":")
    n, band = int(n), int(band)
    pr = None
    for mode in a.modes.split(","):
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(mode_env(mode))
        s = bench.HipSolver(args, 0, 0, n, 2 * n, 10, 1234, 0, 1e-3 if dtype == "f32" else 1e-4, dtype=dtype, band=band or None, pr=pr)
        pr = s.pr
        s.begin(); s.steps(10)
        st0 = s.stats(); s.profiling(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.steps(a.iters)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        st1 = s.stats(); s.profiling(False); kn = s.spmv_kernels(); ri = s.reorder_info(); s.end(); s.close()
        nl, ms = st1["spmv_launches"] - st0["spmv_launches"], st1["spmv_ms"] - st0["spmv_ms"]
        cg = st1["cg_iters"] - st0["cg_iters"]
        avg = ms / nl * 1e-3 if nl else float("nan")
        bps = st1["spmv_bytes"] / 2.0
        print(json.dumps(dict(n=n, dtype=dtype, band=band, mode=mode, spmv_avg_us=avg * 1e6, frac_of_8TBs=bps / avg / 1e9 / 8000.0 if nl else None,
                              us_per_cg_iter=1e6 * el / cg if cg else None, cg_its_per_admm_iter=cg / float(a.iters), launches_timed=int(nl),
                              scs_init_s=s.t_init, kernels=kn, numbering=ri)), flush=True)
