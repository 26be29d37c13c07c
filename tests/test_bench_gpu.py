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
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_one_gpu_and_report_one_line():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--n", "20000",
                          "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--batch-n", "4000",
                          "--batch-per-gpu", "2", "--batch-concurrency", "2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == [0, 1]
    assert d["status"] == "solved" and d["value"] > 0 and d["steps"] == 10
    assert len(d["results_per_rank"]) == 2 and all(r[1] > 0 for r in d["results_per_rank"])
    assert d["results_per_rank"][0][2] != d["results_per_rank"][1][2]  # seed + rank: two different problems
    assert d["batch"]["problems"] == 4 and d["batch"]["all_solved"]
    assert "share_gpu" in d
