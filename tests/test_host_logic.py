"""Host-side logic on CPU: generators, batching over gloo (world_size 2)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from scs_amd import batch, capi, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generator_is_feasible_and_complementary():
    cone = dict(z=5, l=9, bl=[-1.0, -2.0], bu=[0.5, 1.0], q=[1, 2, 7, 30], s=[1, 3, 6])
    m = capi.cone_rows(cone)
    pr = problems.random_cone_prob(25, m, 4, cone, seed=3)
    y, s = pr["y_opt"], pr["s_opt"]
    assert abs(y @ s) <= 1e-9 * max(1.0, np.abs(y).max() * np.abs(s).max())   # s'y = 0
    assert np.abs(problems.proj_dual_cone_np(y, cone) - y).max() <= 1e-9       # y in K*
    assert np.abs(problems.proj_dual_cone_np(-s, cone)).max() <= 1e-9          # s in K  (Moreau)
    A = pr["A"]
    assert np.abs(A @ pr["x_opt"] + s - pr["b"]).max() <= 1e-12 * max(1, np.abs(pr["b"]).max())
    assert np.abs(A.T @ y + pr["c"]).max() <= 1e-12 * max(1, np.abs(pr["c"]).max())
    assert np.all(np.diff(A.indptr) == 4)
    for j in range(A.shape[1]):
        r = A.indices[A.indptr[j]:A.indptr[j + 1]]
        assert np.all(np.diff(r) > 0)


def test_socp_cone_recipe_matches_reference_law():
    c = problems.socp_cone_sizes(4000)   # test/random_socp_prob.c:79-107 with m = 4000
    assert c["z"] == 400 and c["l"] == 1200
    assert sum(c["q"]) == 4000 - 1600
    import math
    assert max(c["q"]) <= math.ceil(4000 / math.log(4000))
    c2 = problems.socp_cone_sizes(1000, q_fixed=8)
    assert set(c2["q"][:-1]) == {8}


def test_partition_is_a_partition():
    for count, world in ((64, 8), (7, 2), (3, 4), (0, 2)):
        seen = sorted(i for r in range(world) for i in batch.partition(count, world, r))
        assert seen == list(range(count))
        sizes = [len(batch.partition(count, world, r)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_batch_roundtrip():
    def solve_one(j, d):
        return dict(status_val=1, iter=25 * j, pobj=float(j), dobj=float(j), res_pri=0.0, res_dual=0.0, gap=0.0)
    tab = batch.run_batch(dict(n=10, m=20, col_nnz=2, seed=1, count=5, aa=0, max_iters=10), solve_one)
    assert tab.shape == (5, len(batch.REC_FIELDS))
    assert list(tab[:, 0]) == [0, 1, 2, 3, 4] and list(tab[:, 2]) == [0, 25, 50, 75, 100]


def test_two_rank_gloo_batch_matches_serial():
    """world_size 2 over gloo on CPU: descriptor broadcast + record all-gather; the solves are
    the oracle restatement (checker used as a stand-in solver in this CPU test only)."""
    script = textwrap.dedent('''
        import os, sys, json
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from scs_amd import batch, capi, problems
        from oracle import pyoracle
        dist.init_process_group(backend="gloo")
        def solve_one(j, d):
            pr = problems.random_socp(d["n"], d["m"], d["col_nnz"], seed=d["seed"] + j)
            prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
            return pyoracle.oracle_solve(prob, max_iters=d["max_iters"])["info"]
        desc = dict(n=40, m=120, col_nnz=4, seed=100, count=5, aa=0, max_iters=400) if dist.get_rank() == 0 else {}
        tab = batch.run_batch(desc, solve_one, dist=dist, device="cpu")
        if dist.get_rank() == 0:
            print("TABLE " + json.dumps(tab.tolist()))
        dist.barrier()
        dist.destroy_process_group()
    ''' % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    # torch.distributed.run needs a file: write one
    import tempfile, json
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(script)
        path = f.name
    try:
        out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                       "--master-addr", "127.0.0.1", "--master-port", "29611", path], env=env, text=True,
                                      stderr=subprocess.STDOUT, timeout=600)
    finally:
        os.unlink(path)
    line = [l for l in out.splitlines() if l.startswith("TABLE ")][0]
    tab = np.array(json.loads(line[6:]))
    assert tab.shape == (5, len(batch.REC_FIELDS))
    from oracle import pyoracle
    for j in range(5):
        pr = problems.random_socp(40, 120, 4, seed=100 + j)
        prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
        info = pyoracle.oracle_solve(prob, max_iters=400)["info"]
        assert int(tab[j, 0]) == j and int(tab[j, 2]) == info["iter"]
        assert abs(tab[j, 3] - info["pobj"]) <= 1e-12 * max(1, abs(info["pobj"]))


def test_bench_cpu_baseline_worker_runs_without_a_gpu():
    """bench.py's cpu_baseline legs are CPU-only child processes (the reference build on the host
    cores): they must work where there is no GPU and print one JSON object each."""
    import json
    import subprocess
    import sys
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # spec = kind:threads:n:col_nnz:seed:aa:q_fixed:i0:k[:numa side]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-baseline-worker", "socp:1:1500:10:1234:0:0:5:10"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["threads"] == 1 and d["n"] == 1500 and d["iter"] == 15  # ONE capped run of i0 + k iterations
    # the window legs run through the counting shim (oracle/trace_linsys.c) so that they know the reference's own CG iterations
    assert d["solve_s"] > 0 and d["flavour"] == "libscsindir_ref_trace.so", d
    # the window [5, 15) is read off the reference's own per-iteration log
    assert d["window"] == [5, 15] and 0 < d["window_s"] < d["solve_s"] and abs(d["its_per_s"] - 10 / d["window_s"]) < 1e-9
    # one row per iteration + the final row, every parity column; the reference's own CG counts per ADMM iteration
    assert [r["iter"] for r in d["log_rows"]] == list(range(15)) + [15]
    assert all(set(r) >= {"res_pri", "res_dual", "gap", "pobj", "dobj"} for r in d["log_rows"])
    assert len(d["cg_its_by_iter"]) == 15 and d["cg_its_window"] == sum(d["cg_its_by_iter"][5:15]) > 0
    # the same run WITHOUT the shim gives the same trajectory: the shim only counts (same last row as a plain logged run)
    from scs_amd import capi, problems
    import tempfile
    ref = pyoracle.load_ref("libscsindir_ref.so")
    pr = problems.random_socp(1500, 3000, 10, seed=1234)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
    with tempfile.TemporaryDirectory() as td:
        log = os.path.join(td, "l.csv")
        capi.solve(ref, prob, verbose=0, acceleration_lookback=0, max_iters=15, log_csv_filename=log.encode())
        last = open(log).read().splitlines()[-1].split(",")
        names = open(log).read().splitlines()[0].split(",")
    assert float(last[names.index("res_pri")]) == d["log_rows"][-1]["res_pri"]
    assert float(last[names.index("pobj")]) == d["log_rows"][-1]["pobj"]


def test_bench_cpu_termination_worker_and_numa_split():
    """the to-termination leg behind batch.parity (kind "term") and the NUMA helper: CPU only."""
    import json
    import subprocess
    import sys
    from oracle import pyoracle
    if not pyoracle.ref_available("libscsindir_ref_omp.so"):
        pytest.skip("oracle/_ref not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-baseline-worker", "term:2:800:10:1000:0:0:0:0:b"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["flavour"] == "libscsindir_ref_omp.so" and d["threads"] == 2 and d["info"]["status_val"] == 1
    assert d["info"]["iter"] % 25 == 0 and d["info"]["iter"] > 0
    sys.path.insert(0, root)
    import bench
    nodes = bench._numa_cpus()
    assert nodes and all(nodes) and sorted(c for nd in nodes for c in nd) == sorted(set(c for nd in nodes for c in nd))


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` outside torchrun re-executes itself as 2 ranks (the launch path the
    driver's scaling run depends on): same rendezvous, barriers, reductions and JSON assembly as on
    GPUs, over gloo with the stub solver (no GPU in this container)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-solver", "--backend", "gloo",
                          "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                         timeout=300, env=env)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == [0, 1]
    assert d["steps"] == 3 and d["warmup"] == 1
    assert len(d["results_per_rank"]) == 2 and len(d["per_rank_it_per_s"]) == 2
    iters = [int(r[1]) for r in d["results_per_rank"]]
    assert iters == [50, 75]  # the stub's ranks converge at different iterations
    # value = all ranks' iterations / the slowest rank's wall time; the window figure is separate
    assert d["value"] > 0 and d["window_it_per_s"] > 0 and abs(d["ms_per_step"] - 2.0) < 2.0


def test_bench_eight_rank_dry_run_over_gloo():
    """the first 8-GPU driver run must not fail on plumbing.  `--gpus 8` over gloo with the stub solver: eight
    ranks rendezvous, the census sees all of them, the configs[3] batch is 64 problems split 8 per rank and every record comes
    back through the all-gather, `value` = sum of the ranks' iterations / the slowest rank's wall time, and ONLY rank 0 writes to
    stdout -- exactly one line (include/scs.h:271-324: scs_init / scs_solve are re-entrant, which is what makes the split legal)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--stub-solver", "--backend", "gloo",
                          "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # ranks 1..7 print nothing on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_ranks_seen"] == list(range(8)) and d["collective_backend"] == "gloo"
    assert d["scaling"] == "weak" and d["config"]["problems_per_gpu"] == 1
    assert len(d["results_per_rank"]) == 8 and len(d["per_rank_it_per_s"]) == 8
    iters = [int(r[1]) for r in d["results_per_rank"]]
    assert iters == [25 * (2 + r) for r in range(8)]  # the stub's rank r converges at iteration 25 (2 + r)
    # value = all ranks' iterations / max-over-ranks wall: the slowest rank (225 iterations of 2 ms) bounds the wall from below
    wall = sum(iters) / d["value"]
    assert 0.002 * 225 <= wall <= 0.002 * 225 * 3 + 2.0, wall
    assert d["ms_per_iter_whole_solve"] == pytest.approx(1000.0 / d["value"], rel=1e-3)
    b = d["batch"]
    assert b["problems"] == 64 and b["problems_per_gpu"] == 8 and b["ranks"] == 8 and b["all_solved"] is True
    assert b["iters_sum"] == sum(25 * (1 + j % 3) for j in range(64))  # every problem's record reached rank 0 exactly once
    assert b["admm_iters_per_s"] == pytest.approx(b["iters_sum"] / b["wall_s"], rel=1e-3)
    assert len(lines[0]) < 4096


def test_bench_line_is_compact_and_carries_the_contract_keys(tmp_path):
    """the driver could not parse a 21 KB line.  The LAST stdout line must be one compact JSON object
    (< 4096 bytes) with the contract's keys; the full record goes to `detail_file` and, prefixed, to stderr.  Checked twice:
    on the stub launch path end to end, and on round 4's full 21 KB record (the worst case the builder has produced)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SCS_BENCH_DETAIL"] = str(tmp_path / "detail.json")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--stub-solver", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_MAX_BYTES
    d = json.loads(lines[0])
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline", "detail_file"}
    assert need <= set(d), need - set(d)
    # `value` (whole solve) and `ms_per_step` (the timed window) describe different regions -- the line says so itself
    assert {"value_definition", "ms_per_iter_whole_solve", "ms_per_step_definition"} <= set(d)
    assert len(d["value_definition"]) <= 120 and d["ms_per_iter_whole_solve"] == pytest.approx(1000.0 / d["value"], rel=1e-3)
    assert d["roofline"]["kernel"].startswith("stub")  # the label comes from the solver (scs_amd_get_spmv_kernel_name), not from a constant
    assert {"workload", "n", "m", "nnz"} <= set(d["config"]) and {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"} <= set(d["roofline"])
    assert json.loads(json.dumps(d)) == d
    full = json.load(open(d["detail_file"]))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and "results_per_rank" in full
    det = [l for l in out.stderr.splitlines() if l.startswith(bench.DETAIL_PREFIX)]
    assert len(det) == 1 and json.loads(det[0][len(bench.DETAIL_PREFIX):])["steps"] == 3
    big = json.load(open(os.path.join(root, "profiles", "r4_bench_default_v6.json")))  # the line that broke the parser
    c = bench.compact_line(big, "/x/detail.json")
    s = json.dumps(c, separators=(",", ":"))
    assert len(s) < bench.LINE_MAX_BYTES and need <= set(c)
    assert {"value", "unit", "cores", "kind", "sample", "host_cores", "cpu_model"} <= set(c["cpu_baseline"]) and len(c["cpu_baseline"]["sample"]) <= 200
    assert c["roofline"]["frac"] == pytest.approx(big["roofline"]["frac"], rel=1e-5)
    assert c["parity_window"]["max_rel_diff"] == pytest.approx(big["parity_window"]["max_rel_diff"], rel=1e-5)
    assert c["batch"]["parity"]["ok"] is True and c["secondary"]["configs2_sdp"]["ms_per_projection"] > 0
    # round 6 keys travel through the compaction: the many-small-SOC variant, the spread of iters_to_eps
    big["secondary"]["headline_many_small_soc"] = dict(status="solved", q=8, soc_cones=150000, iters_to_eps=500, time_to_eps_s=9.0, value_it_per_s=55.5,
                                                      cone_us_per_projection=20.0, workload="x" * 300)
    big["iters_to_eps_spread"] = dict(observed=[475, 525, 575], reference_cpu=575, this_run=525, step=25, source="s")
    big["roofline"]["kernel_short"] = "csr_wave_lockstep_kernel<EPI,16,4> (CSR SpMV, A and A')"
    c = bench.compact_line(big, "/x/detail.json")
    assert c["secondary"]["headline_many_small_soc"] == dict(status="solved", q=8, soc_cones=150000, iters_to_eps=500, time_to_eps_s=9.0,
                                                             value_it_per_s=55.5, cone_us_per_projection=20.0)
    assert c["iters_to_eps_spread"]["observed"] == [475, 525, 575] and c["roofline"]["kernel"].startswith("csr_wave_lockstep_kernel<EPI,16,4>")
    assert len(json.dumps(c, separators=(",", ":"))) < bench.LINE_MAX_BYTES
    # an absurdly inflated record still yields a parseable line: optional blocks are shed, the contract keys stay
    big["config"]["workload"] = "x" * 5000
    big["secondary"]["locality_variant"] = {"k%d" % i: dict(roofline=dict(frac=0.1)) for i in range(400)}
    c = bench.compact_line(big, "/x/detail.json")
    assert len(json.dumps(c, separators=(",", ":"))) < bench.LINE_MAX_BYTES and need <= set(c)


def test_row_slabs_of_the_sharded_linear_system_cover_every_row_once():
    from scs_amd import shard
    for m, world in ((15001, 2), (10, 3), (7, 8), (1000000, 8)):
        slabs = [shard.slab(m, world, r) for r in range(world)]
        assert slabs[0][0] == 0 and slabs[-1][1] == m
        assert all(a[1] == b[0] for a, b in zip(slabs[:-1], slabs[1:]))
        sizes = [b - a for a, b in slabs]
        assert max(sizes) - min(sizes) <= 1


def test_row_sharded_pcg_over_two_gloo_ranks():
    """world_size 2 over gloo on CPU: the data-path collective of the row-sharded linear solve (one all-reduce of an n-vector
    per CG iteration) and its PCG logic, the slab products played by scipy; checked against a sparse direct solve of the KKT system"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29641", os.path.join(root, "tests", "shard_cpu_worker.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("SHARDCPU ")][0][9:])
    assert d["world"] == 2 and d["rows"] == [0, 401]
    assert d["cg_iters"] > 10 and d["allreduce_calls"] >= d["cg_iters"]
    assert d["err"] <= 1e-9, d


def test_parallel_host_transpose_of_the_b1_boundary_equals_the_serial_one(tmp_path):
    """Round 6: scs_init_lin_sys_work's CSC -> CSR transpose (host arrays at the B1 boundary; linsys/cpu/indirect/private.c:7-46) runs on four
    threads from a few million entries on (the nnz = 2.2e9 run spent two minutes in the serial loop).  Pinned on the CPU: byte-identical
    to the serial counting sort for ragged / empty columns, 32- and 64-bit entry positions, fp64 and fp32 (scs_amd/csrc/host_transpose.h)."""
    import ctypes
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so = str(tmp_path / "libtr.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-shared", "-fPIC", "-o", so, os.path.join(here, "native", "host_check_transpose.cpp")])
    assert ctypes.CDLL(so).transpose_check() == 0
