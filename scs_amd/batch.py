"""Batched solves of independent problems across the GPUs of one node.

The SCS path has no collective inside a solve: problems are
independent, so problem j goes to rank j % world, each rank (one process per GPU)
runs its queue through the C ABI, and RCCL over xGMI (torch.distributed backend
"nccl") carries only (i) the broadcast batch descriptor and (ii) the all-gather of
fixed-size result records.  The same code runs over gloo on CPU in the tests with
an injected solve function.
"""
import numpy as np

REC_FIELDS = ("index", "status_val", "iter", "pobj", "dobj", "res_pri", "res_dual", "gap", "solve_time_ms")


def partition(num_problems, world, rank):
    """Indices owned by `rank`: round-robin, so queues differ by at most one problem."""
    return list(range(rank, num_problems, world))


def broadcast_descriptor(desc, dist, device):
    """desc: dict of ints on rank 0 (ignored elsewhere) -> same dict on every rank."""
    import torch
    keys = ("n", "m", "col_nnz", "seed", "count", "aa", "max_iters")
    t = torch.zeros(len(keys), dtype=torch.int64, device=device)
    if dist is None or dist.get_rank() == 0:
        t = torch.tensor([int(desc[k]) for k in keys], dtype=torch.int64, device=device)
    if dist is not None:
        dist.broadcast(t, src=0)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def gather_records(local_records, num_problems, dist, device):
    """local_records: list of tuples following REC_FIELDS.  Returns (num_problems, len(REC_FIELDS))
    float64 array ordered by problem index, identical on every rank."""
    import torch
    world = 1 if dist is None else dist.get_world_size()
    per_rank = (num_problems + world - 1) // world
    buf = torch.full((per_rank, len(REC_FIELDS)), float("nan"), dtype=torch.float64, device=device)
    for i, rec in enumerate(local_records):
        buf[i] = torch.tensor([float(v) for v in rec], dtype=torch.float64, device=device)
    if dist is None:
        allb = [buf]
    else:
        allb = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(allb, buf)
    out = np.full((num_problems, len(REC_FIELDS)), np.nan)
    for b in allb:
        for row in b.cpu().numpy():
            if not np.isnan(row[0]):
                out[int(row[0])] = row
    return out


def run_batch(desc, solve_one, dist=None, device="cpu"):
    """desc: batch descriptor (rank 0's is authoritative).  solve_one(index, desc) -> info dict with
    the ScsInfo fields.  Returns the gathered record table."""
    d = broadcast_descriptor(desc, dist, device)
    rank = 0 if dist is None else dist.get_rank()
    world = 1 if dist is None else dist.get_world_size()
    recs = []
    for j in partition(d["count"], world, rank):
        info = solve_one(j, d)
        recs.append((j, info["status_val"], info["iter"], info["pobj"], info["dobj"], info["res_pri"],
                     info["res_dual"], info["gap"], info.get("solve_time", float("nan"))))
    return gather_records(recs, d["count"], dist, device)
