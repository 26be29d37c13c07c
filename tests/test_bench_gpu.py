"""bench.py's N-rank path with the real solver: two ranks sharing the one GPU of a test box (`--share-gpu`: rank r on
GPU r % visible GPUs, collectives over gloo because RCCL refuses two ranks on one device).  Exercises the self-spawn
under torch.distributed.run, the descriptor broadcast, the rank census, per-rank solves, the result all-gather and
the configs[3] batch workload -- everything of the multi-GPU run except RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
# Per-iteration bounds on the two per-iteration logs (same problem, same iterations, same logged tolerance schedule; worst column
# of each row).  Row 0: the first linear solve runs to the 1e-12 floor on both sides -> rounding only.  Row 1: one
# inexact solve apart (O(CG tolerance) of a residual that is still large).  From row 2 on two valid inexact trajectories separate
# (DESIGN.md section 4).  Measured on the GPU box at this test's size (n = 1.2e5, gpurun_out/r5a): 9.0e-13, 9.0e-4, 0.146, 0.146
# (the last row repeats the returned state) -- the bounds are those figures x 2 (row 0: the 1e-10 asked for).  At the headline size
# the same rows measure 3.6e-14, 6.7e-5, 6.8e-3, 0.215 (bench line, parity_window.max_rel_diff_by_iter).
# Round 6: at this size scs_init now renumbers the problem (chain + home, reorder.cpp) -- another summation order in every product, so
# row 1 (O(CG tolerance)) moved to 2.2e-3 (gpurun_out/r6tests); its bound is that x 2.
PARITY_WINDOW_BOUNDS = [1e-10, 4.4e-3, 0.3, 0.3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line_and_detail(out):
    """stdout = ONE compact JSON line (what the driver parses, < 4 KB); the full record it summarises is in `detail_file`.
    Returns the full record after checking the line against it."""
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.strip().splitlines()[-1] == lines[0], out.stdout[-2000:]
    assert len(lines[0]) < 4096, len(lines[0])
    c = json.loads(lines[0])
    d = json.load(open(c["detail_file"]))
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "dtype", "data", "scaling", "higher_is_better", "status"):
        assert c[k] == d[k], k
    assert abs(c["value"] - d["value"]) <= 1e-5 * abs(d["value"]) and abs(c["ms_per_step"] - d["ms_per_step"]) <= 1e-5 * d["ms_per_step"]
    assert c["config"]["n"] == d["config"]["n"] and "roofline" in c and "cpu_baseline" in c
    return d


def test_two_ranks_share_one_gpu_and_report_one_line():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--n", "20000",
                          "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--batch-n", "4000",
                          "--batch-per-gpu", "2", "--batch-concurrency", "2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _line_and_detail(out)
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == [0, 1]
    assert d["status"] == "solved" and d["value"] > 0 and d["steps"] == 10
    assert len(d["results_per_rank"]) == 2 and all(r[1] > 0 for r in d["results_per_rank"])
    assert d["results_per_rank"][0][2] != d["results_per_rank"][1][2]  # seed + rank: two different problems
    assert d["batch"]["problems"] == 4 and d["batch"]["all_solved"]
    assert "share_gpu" in d


def _run_bench(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MASTER_ADDR"] = "127.0.0.1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                          "--secondary", "batch", "--batch-n", "20000", "--batch-per-gpu", "2", "--batch-concurrency", "2"] + extra,
                         env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return _line_and_detail(out)


def test_one_rank_under_torchrun_runs_its_collectives_over_rccl():
    """the first 8-GPU run must not be the first RCCL run.  `--torchrun` sends the N=1 bench through
    torch.distributed.run (WORLD_SIZE=1): init_process_group("nccl") (= RCCL on ROCm), the rank census all_gather, the
    descriptor broadcast, the MAX / SUM all_reduces, the result all_gather, the barriers and the configs[3] batch
    (broadcast_descriptor + gather_records) all run on cuda tensors.  Same headline problem as the plain N=1 run: the two
    whole-solve rates agree (bound 25 %, measured within 5 %)."""
    plain = _run_bench(["--gpus", "1"])
    rccl = _run_bench(["--gpus", "1", "--torchrun", "--backend", "nccl"])
    assert plain["collective_backend"] is None and rccl["collective_backend"] == "nccl"
    assert rccl["n_gpus"] == 1 and rccl["rccl_ranks_seen"] == [0]
    assert rccl["status"] == plain["status"] == "solved"
    assert rccl["iters_to_eps"] == plain["iters_to_eps"]            # bit-reproducible solve, same seed
    assert rccl["final"]["pobj"] == plain["final"]["pobj"]
    # (same work, two processes: the rates agree to the run-to-run spread of a 1.5 s solve sharing the box's host with the other test
    # processes -- measured up to 5 %; the point of this test is the collectives, the iterations and the objective above, so the bound
    # on the RATE is kept wide: a timing assertion must not be what fails a parity suite on a busy box)
    assert abs(rccl["value"] - plain["value"]) <= 0.25 * plain["value"], (rccl["value"], plain["value"])
    assert len(rccl["per_rank_it_per_s"]) == 1 and rccl["per_rank_it_per_s"][0] > 0
    assert rccl["batch"]["problems"] == 2 and rccl["batch"]["all_solved"]


def test_side_workloads_of_the_default_line_at_reduced_sizes():
    """The `secondary` block of the default bench line (configs[2] SDP, configs[4] fp32 carried to eps with fp64 host
    residuals, the locality variants) and the OpenMP-sweep plumbing, at sizes that take seconds: every field the driver's
    line carries must be there and sane."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "120000", "--fp32-n", "120000", "--steps", "10", "--warmup", "5",
                          "--batch-n", "8000", "--batch-per-gpu", "2", "--batch-concurrency", "2", "--cpu-omp-sweep", "4", "--cpu-omp-budget", "120",
                          "--cpu-window-iters", "2", "--parity-threads", "4", "--aa-window-threads", "4"], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _line_and_detail(out)
    assert d["status"] == "solved" and d["roofline"]["bound"] == "hbm" and "parity_mode" in d
    s = d["secondary"]
    assert s["configs2_sdp"]["status"] == "solved" and s["configs2_sdp"]["ms_per_projection"] > 0
    big = {(c["order"], c["blocks"]): c for c in s["psd_large_blocks"]["cases"]}  # PSD blocks beyond the LDS path (fused step)
    assert set(big) == {(92, 64), (256, 8), (1024, 1)}
    assert all(c["ms_per_projection"] > 0 and c["projections_timed"] > 0 and not c["psd_unconverged"] for c in big.values())
    assert big[(92, 64)]["ms_per_projection"] < 3.0 and big[(256, 8)]["ms_per_projection"] < 4.0 and big[(1024, 1)]["ms_per_projection"] < 16.0
    f32 = s["configs4_fp32"]
    assert f32["status"] == "solved" and f32["iters_to_eps"] > 0
    rec = f32["final_fp64_host_recomputed"]  # the fp32 solver's own stopping test, re-done in fp64: within rounding of its limits
    for k in ("res_pri", "res_dual", "gap"):
        assert rec[k] <= 1.5 * rec["limits"][k], (k, rec)
    for band in ("band_1024", "band_4096"):
        assert s["locality_variant"][band]["window_it_per_s"] > 0
    lv = s["locality_variant"]  # the scrambled band: renumbered inside scs_init (reorder.h) vs solved as given
    assert lv["permuted_band_1024"]["numbering"]["renumbered"] is True and lv["permuted_band_1024_as_given"]["numbering"]["renumbered"] is False
    used, given = lv["permuted_band_1024"]["numbering"]["lines_per_entry_used"], lv["permuted_band_1024"]["numbering"]["lines_per_entry_given"]
    assert sum(used) < 0.5 * sum(given), (used, given)
    assert d["batch"]["all_solved"]
    aa = s["headline_aa_on"]  # the reference's default settings (acceleration_lookback=10) on the headline problem
    assert aa["status"] == "solved" and aa["iters_to_eps"] > 0 and aa["value_it_per_s"] > 0 and aa["accel_time_s"] >= 0
    assert aa["accepted_accel_steps"] + aa["rejected_accel_steps"] > 0
    from oracle import pyoracle
    if pyoracle.ref_available():
        cb = d["cpu_baseline"]
        assert cb["value"] > 0 and cb["cores"] == 1
        # the window is priced in the reference's OWN CG iterations (counting shim), both sides on the logged schedule
        assert cb["cpu_cg_its_window"] > 0 and cb["gpu_cg_its_window"] > 0 and cb["cpu_s_per_cg_iter"] > 0
        assert cb["cpu_time_to_eps_s_estimate"] > 0 and cb["gpu_over_cpu_same_window"] > 0
        # the same schedule on both sides: the two CG counts over the window agree closely (not exactly: summation order)
        assert abs(cb["cpu_cg_its_window"] - cb["gpu_cg_its_window"]) <= 0.2 * cb["cpu_cg_its_window"], cb
        legs = d["cpu_baseline_omp"]["legs"]
        assert legs and legs[0]["cores"] == 4 and legs[0]["value"] > 0
        pw = d["parity_window"]  # same problem, same iterations, same (logged) schedule: every row of the two logs side by side
        assert pw["rows"] == 4 and len(pw["rel_diff_per_iter"]) == 4 and [r["iter"] for r in pw["gpu"]] == [0, 1, 2, 3]
        # per-iteration bounds: row j = the state after iteration j, worst column of the row
        worst = [max(v for k, v in r.items() if k != "iter" and isinstance(v, float)) for r in pw["rel_diff_per_iter"]]
        print("parity_window worst-per-row:", worst)
        for j, (w, bound) in enumerate(zip(worst, PARITY_WINDOW_BOUNDS)):
            assert w <= bound, (j, w, bound, pw["rel_diff_per_iter"])
        bp = d["batch"]["parity"]  # one configs[3]-shaped problem to TERMINATION on both sides
        assert bp["ok"] is True and bp["same_status"] and bp["ours_verify"]["ok"], bp
        assert 0.5 <= bp["iter_ratio"] <= 2.0 and bp["pobj_rel_diff"] <= 1e-3
        ar = aa["cpu_reference"]
        assert ar["its_per_s"] > 0 and ar["window"] == [1, 22]
        # (no AA decision falls into the window: the first solve needs acceleration_lookback = 10 samples, one per call, one call
        # every acceleration_interval = 10 iterations -> iteration 100, src/aa.c; the GPU side above runs the whole solve)
        assert ar["accepted_accel_steps"] + ar["rejected_accel_steps"] == 0
