"""Intra-problem sharding, functional: ONE linear system split by rows of A across ranks
(scs_amd/shard.py) -- each rank applies its slab's term of G with the MI355X SpMV kernels through the device-pointer
entries of the C ABI, one all-reduce of an n-vector per CG iteration forms G p.
 * two ranks sharing the one GPU of a test box, collectives over gloo (RCCL refuses two ranks on one device);
 * one rank with backend nccl: the SAME data-path all-reduce executes over RCCL on cuda tensors.
Checked against the unsplit solve of libscsamd_linsys.so and against the reference backend."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, backend, port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py"),
                          backend, "6000", "15001", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD ")]
    assert len(line) == 1, out.stdout[-2000:]
    return json.loads(line[0][6:])


@pytest.mark.parametrize("world,backend,port", [(2, "gloo", 29631), (1, "nccl", 29632)])
def test_row_sharded_linear_solve_matches_the_unsplit_one(world, backend, port):
    d = _run(world, backend, port)
    assert d["world"] == world and d["backend"] == backend
    assert d["cg_iters"] > 20 and d["allreduce_calls"] >= d["cg_iters"]  # one n-vector all-reduce per CG iteration (+ warm start)
    assert d["err_x"] <= 1e-8 and d["err_y"] <= 1e-8 and d["err_x2"] <= 1e-8, d
    if "err_vs_reference" in d:
        assert d["err_vs_reference"] <= 1e-7, d
