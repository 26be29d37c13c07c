#!/usr/bin/env python3
"""bench.py -- ADMM iterations/sec of the MI355X-native SCS hot path.

Metric (BASELINE.json): "ADMM iters/sec + time-to-eps=1e-4, 1e6-var random SOCP,
1 GPU".  Workload at N=1: BASELINE configs[1] -- random SOCP n=1e6, m=2e6,
nnz=1e7 (col_nnz=10), zero + nonnegative + second-order cones, fp64, indirect
(PCG) linear solves, default SCS settings except acceleration_lookback=0 (on this
problem family the reference's own safeguard rejects every Anderson step; the AA-on
figures are in DESIGN.md section 7).

A "step" is ONE ADMM iteration (linear-system solve by PCG + cone projection +
the vector glue) on inputs resident in HBM.  The run does W untimed warm-up
iterations, then times EXACTLY K iterations between barrier+synchronize pairs,
then (N=1) keeps iterating to eps=1e-4 to report time-to-eps.  With N>1 each rank
solves its own independent problem of the same size (weak scaling; the path has
no intra-solve collective -- RCCL only carries the batch descriptor and the
result records).

One JSON line on stdout (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--m", type=int, default=0, help="default 2n")
    ap.add_argument("--col-nnz", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--aa", type=int, default=0, help="acceleration_lookback (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-to-eps", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not sample SpMV launches with HIP events in the timed region (roofline left empty)")
    ap.add_argument("--q-fixed", type=int, default=0,
                    help="many-small-cones variant of the same workload: every SOC has this size (SURVEY 8d)")
    ap.add_argument("--cpu-sample-n", type=int, default=30000)
    ap.add_argument("--cpu-omp-sample-n", type=int, default=30000,
                    help="sample size for the OpenMP flavour of the reference (0 = skip it)")
    ap.add_argument("--cpu-omp-threads", type=int, default=0, help="default min(nproc, 16)")
    ap.add_argument("--cpu-baseline-timeout", type=float, default=150.0,
                    help="hard wall-clock cap (s) for each CPU baseline sample; it runs in a child process")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sample-i0", type=int, default=20)
    ap.add_argument("--cpu-sample-iters", type=int, default=25)
    ap.add_argument("--max-iters", type=int, default=20000)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="f32 = the -DSFLOAT library (BASELINE configs[4]; residual tolerance relaxed to 1e-3)")
    ap.add_argument("--eps", type=float, default=0.0, help="eps_abs = eps_rel (default 1e-4; 1e-3 with --dtype f32)")
    return ap.parse_args()


def cpu_baseline(args, full_n, threads=1):
    """The reference's own CPU indirect solver (oracle/_ref, built from /root/reference by
    oracle/Makefile) on the host cores, bounded sample: the same generator at a reduced n,
    iterations [i0, i1) isolated by differencing two capped runs (the first iterations solve
    to 1e-12 and are not representative), scaled linearly in nnz to the full size.
    threads == 1: the stock build; threads > 1: the reference's USE_OPENMP flavour."""
    flavour = "libscsindir_ref.so" if threads == 1 else "libscsindir_ref_omp.so"
    try:
        from oracle import pyoracle
        from scs_amd import capi, problems
        if not pyoracle.ref_available():
            return None
        if threads > 1:
            os.environ["OMP_NUM_THREADS"] = str(threads)  # read when libgomp initialises (first load)
            # libgomp's default active spinning makes the reference's many tiny parallel regions
            # 20x slower than serial whenever anything else shares the cores; passive waiting is its best case
            os.environ["OMP_WAIT_POLICY"] = "passive"
        ref = pyoracle.load_ref(flavour)
        n = min(args.cpu_sample_n if threads == 1 else args.cpu_omp_sample_n, full_n)
        pr = problems.random_socp(n, 2 * n, args.col_nnz, seed=args.seed, q_fixed=args.q_fixed or None)
        prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
        i0, i1 = args.cpu_sample_i0, args.cpu_sample_i0 + args.cpu_sample_iters
        t0 = time.time()
        ra = capi.solve(ref, prob, verbose=0, acceleration_lookback=args.aa, max_iters=i0)["info"]
        rb = capi.solve(ref, prob, verbose=0, acceleration_lookback=args.aa, max_iters=i1)["info"]
        wall = time.time() - t0
        dt = (rb["solve_time"] - ra["solve_time"]) / 1e3
        its_per_s = (rb["iter"] - ra["iter"]) / dt
        scale = n / float(full_n)
        return dict(value=its_per_s * scale, unit="ADMM iters/sec", cores=threads, kind="reference",
                    sample=(f"reference {flavour} (linsys/cpu/indirect, {threads} thread(s)) on the same generator at "
                            f"n={n}, m={2*n}, nnz={n*args.col_nnz}: ADMM iterations {ra['iter']}..{rb['iter']} in "
                            f"{dt:.2f} s = {its_per_s:.3f} it/s (difference of two capped runs), scaled by "
                            f"n_sample/n_full={scale:g} (cost per iteration is linear in nnz)"),
                    measured_it_per_s=its_per_s, sample_wall_s=wall, host_cores=os.cpu_count())
    except Exception as e:  # the baseline is reported, never required
        return dict(value=None, unit="ADMM iters/sec", cores=threads, kind="reference", sample=f"unavailable: {e}")


def cpu_baseline_bounded(args, full_n, threads):
    """cpu_baseline() in a child process with a hard timeout, so that a slow or oversubscribed
    host can never stall the bench line (the baseline is reported, never required)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), "--n", str(full_n),
           "--col-nnz", str(args.col_nnz), "--seed", str(args.seed), "--aa", str(args.aa),
           "--q-fixed", str(args.q_fixed), "--cpu-sample-n", str(args.cpu_sample_n),
           "--cpu-omp-sample-n", str(args.cpu_omp_sample_n), "--cpu-sample-i0", str(args.cpu_sample_i0),
           "--cpu-sample-iters", str(args.cpu_sample_iters)]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                           timeout=args.cpu_baseline_timeout)
        return json.loads(p.stdout.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="ADMM iters/sec", cores=threads, kind="reference",
                    sample=f"unavailable: sample exceeded the {args.cpu_baseline_timeout:.0f} s cap on this host")
    except Exception as e:
        return dict(value=None, unit="ADMM iters/sec", cores=threads, kind="reference", sample=f"unavailable: {e}")


def main():
    args = parse()
    if args.cpu_baseline_worker:  # child of cpu_baseline_bounded: CPU only, no GPU, no torch
        print(json.dumps(cpu_baseline(args, args.n, threads=args.cpu_baseline_worker)), flush=True)
        return
    n = args.n
    m = args.m or 2 * n
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group(backend="nccl")  # RCCL
    from scs_amd import capi, problems
    lib = capi.load("libscsamd_f32.so" if args.dtype == "f32" else "libscsamd.so")
    T = lib._scs_types
    assert lib.scs_amd_set_device(local_rank) == 0
    eps = args.eps or (1e-3 if args.dtype == "f32" else 1e-4)

    # ---- batch descriptor: rank 0 decides, RCCL broadcast (launch) -------------
    desc = torch.tensor([n, m, args.col_nnz, args.seed, args.steps, args.warmup, args.aa], dtype=torch.int64,
                        device="cuda")
    if dist:
        dist.broadcast(desc, src=0)
    n, m, col_nnz, seed, K, W, aa = [int(v) for v in desc.tolist()]

    # ---- synthetic problem, one per rank (seed + rank) --------------------------
    t0 = time.time()
    pr = problems.random_socp(n, m, col_nnz, seed=seed + rank, dtype=T.np_float, q_fixed=args.q_fixed or None)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"], T=T)
    t_gen = time.time() - t0
    st = capi.default_settings(lib, verbose=0, acceleration_lookback=aa, max_iters=args.max_iters, eps_abs=eps,
                               eps_rel=eps)
    t0 = time.time()
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    if not w:
        raise SystemExit("scs_init failed")
    t_init = time.time() - t0
    x = np.zeros(n, dtype=T.np_float); y = np.zeros(m, dtype=T.np_float); s = np.zeros(m, dtype=T.np_float)
    sol = T.ScsSolution(x.ctypes.data_as(T.fp), y.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp))
    info = T.ScsInfo()

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up: everything scs_solve does before the loop + W iterations -----
    t_solve0 = time.time()
    assert lib.scs_amd_solve_begin(w, None, 0) == 0
    it = lib.scs_amd_solve_steps(w, W)
    assert it >= 0
    stats0 = T.ScsAmdStats()
    lib.scs_amd_get_stats(w, C.byref(stats0))
    if not args.no_kernel_timing:
        lib.scs_amd_set_profiling(w, 1)  # samples 1 in 8 SpMV launches with HIP events on OUR stream
    # ---- timed region: exactly K ADMM iterations --------------------------------
    barrier()
    t0 = time.perf_counter()
    it2 = lib.scs_amd_solve_steps(w, K)
    barrier()
    elapsed = time.perf_counter() - t0
    assert it2 >= 0
    steps_done = it2 - it
    stats1 = T.ScsAmdStats()
    lib.scs_amd_get_stats(w, C.byref(stats1))
    lib.scs_amd_set_profiling(w, 0)
    t_max = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    steps_t = torch.tensor([steps_done], dtype=torch.int64, device="cuda")
    if dist:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(steps_t, op=dist.ReduceOp.SUM)
    elapsed_max = float(t_max.item())
    total_steps = int(steps_t.item())

    # ---- run on to eps = 1e-4 (time-to-eps), N = 1 only in the JSON -------------
    tte = None
    if not args.no_time_to_eps:
        while not lib.scs_amd_solve_converged(w):
            cur = lib.scs_amd_solve_steps(w, 100)
            if cur < 0 or cur >= args.max_iters:
                break
        torch.cuda.synchronize()
        tte = time.time() - t_solve0
    lib.scs_amd_solve_end(w, C.byref(sol), C.byref(info))
    res = capi.info_dict(info)

    # ---- result records gathered over RCCL (collect) -----------------------------
    rec = torch.tensor([res["status_val"], res["iter"], res["pobj"], res["dobj"], res["res_pri"], res["res_dual"],
                        res["gap"], res["solve_time"]], dtype=torch.float64, device="cuda")
    recs = [rec]
    if dist:
        recs = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(recs, rec)

    if rank == 0:
        cg_its = stats1.cg_iters - stats0.cg_iters
        spmv_samples = stats1.spmv_launches - stats0.spmv_launches
        spmv_ms = stats1.spmv_ms - stats0.spmv_ms
        bytes_per_spmv = stats1.spmv_bytes / 2.0  # already computed with sizeof(scs_float) of the library
        roof = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None,
                    kernel="csr_wave_kernel (wave-owned rows CSR SpMV, both orientations)")
        if spmv_samples > 0 and spmv_ms > 0:
            avg_s = spmv_ms / spmv_samples * 1e-3
            roof["achieved"] = bytes_per_spmv / avg_s / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            # context only: the best pure streaming kernel measured on this chip (lab/spmv_lab.hip) reaches 6.9 TB/s
            roof["frac_of_measured_stream_6900GBs"] = roof["achieved"] / 6900.0
            roof["avg_launch_us"] = avg_s * 1e6
            roof["algorithmic_bytes_per_launch"] = bytes_per_spmv
            roof["launches_timed"] = int(spmv_samples)
        # HBM traffic per launch from the committed PMC passes (rocprofv3 cannot run inside this
        # process); only quoted for the exact workload it was measured on
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            if n == 1000000 and m == 2000000 and col_nnz == 10:
                roof["traffic"] = pj["hbm_bytes_per_launch_mean"]
                roof["traffic_source"] = "profiles/r1_pmc_traffic.json (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)"
        except Exception:
            pass
        out = {
            "metric": "ADMM iters/sec (+ time-to-eps=1e-4), 1e6-var random SOCP, 1 GPU",
            "value": total_steps / elapsed_max,
            "unit": "ADMM iters/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed_max / max(steps_done, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"random SOCP n={n} m={m} nnz={n*col_nnz} (BASELINE configs[1]"
                                   f"{', many-small-cones variant q_i=%d' % args.q_fixed if args.q_fixed else ''}); "
                                   f"cones z={pr['cone']['z']} l={pr['cone']['l']} soc={len(pr['cone']['q'])}; "
                                   f"indirect PCG; acceleration_lookback={aa}",
                       "n": n, "m": m, "nnz": n * col_nnz, "problems_per_gpu": 1,
                       "partition": "one independent problem per GPU"},
            "roofline": roof,
            "cg_its_per_admm_iter": cg_its / max(steps_done, 1),
            "time_to_eps_s": tte,
            "iters_to_eps": res["iter"] if res["status_val"] == 1 else None,
            "status": res["status"],
            "final": {k: res[k] for k in ("pobj", "dobj", "res_pri", "res_dual", "gap", "iter")},
            "setup_s": {"generate": t_gen, "scs_init": t_init},
            "results_per_rank": [[float(v) for v in r.tolist()] for r in recs],
        }
        out["eps"] = eps
        if world == 1 and not args.no_cpu_baseline and args.dtype == "f64":
            out["cpu_baseline"] = cpu_baseline_bounded(args, n, 1)
            if args.cpu_omp_sample_n > 0:  # also the reference's OpenMP flavour on the host cores
                nthr = args.cpu_omp_threads or min(os.cpu_count() or 1, 16)
                out["cpu_baseline_omp"] = cpu_baseline_bounded(args, n, max(2, nthr))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    lib.scs_finish(w)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
