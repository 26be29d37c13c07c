#!/usr/bin/env python3
"""BASELINE configs[3]: a batch of independent SOCPs (default 64 x n=2e5) sharded
round-robin over the GPUs of one node, one process per GPU (torch.distributed / RCCL
carries only the batch descriptor and the result records, scs_amd/batch.py).  Inside a
rank, `--concurrency` host threads each drive their own ScsWork on their own HIP stream,
so small problems overlap their launch gaps.

    python scripts/bench_batch.py --count 8 --n 200000                # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
           scripts/bench_batch.py --count 64 --n 200000               # 8 GPUs
"""
import argparse, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--count", type=int, default=64)
ap.add_argument("--n", type=int, default=200000)
ap.add_argument("--col-nnz", type=int, default=10)
ap.add_argument("--seed", type=int, default=1000)
ap.add_argument("--aa", type=int, default=0)
ap.add_argument("--max-iters", type=int, default=20000)
ap.add_argument("--concurrency", type=int, default=4)
a = ap.parse_args()

import torch
from scs_amd import batch, capi, problems
rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group(backend="nccl")
lib = capi.load("libscsamd.so")
assert lib.scs_amd_set_device(local) == 0
desc = batch.broadcast_descriptor(dict(n=a.n, m=2 * a.n, col_nnz=a.col_nnz, seed=a.seed, count=a.count, aa=a.aa,
                                       max_iters=a.max_iters), dist, "cuda")
mine = batch.partition(desc["count"], world, rank)
probs = {}
for j in mine:  # generation is not part of the timed region
    pr = problems.random_socp(desc["n"], desc["m"], desc["col_nnz"], seed=desc["seed"] + j)
    probs[j] = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])

def solve_one(j):
    r = capi.solve(lib, probs[j], verbose=0, acceleration_lookback=desc["aa"], max_iters=desc["max_iters"])
    i = r["info"]
    return (j, i["status_val"], i["iter"], i["pobj"], i["dobj"], i["res_pri"], i["res_dual"], i["gap"], i["solve_time"] + i["setup_time"])

if dist: dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
with ThreadPoolExecutor(max_workers=max(1, a.concurrency)) as ex:
    recs = list(ex.map(solve_one, mine))
torch.cuda.synchronize()
if dist: dist.barrier()
elapsed = time.perf_counter() - t0
tab = batch.gather_records(recs, desc["count"], dist, "cuda")
tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
if dist: dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
if rank == 0:
    iters = int(np.nansum(tab[:, 2]))
    print(json.dumps(dict(metric="batched independent SOCP solves", n_gpus=world, problems=desc["count"], n=desc["n"],
                          m=desc["m"], concurrency=a.concurrency, wall_s=float(tmax.item()),
                          problems_per_s=desc["count"] / float(tmax.item()), admm_iters_per_s=iters / float(tmax.item()),
                          all_solved=bool(np.all(tab[:, 1] == 1)), iters_min_max=[int(np.nanmin(tab[:, 2])), int(np.nanmax(tab[:, 2]))])))
if dist:
    dist.barrier(); dist.destroy_process_group()
