"""Worker of tests/test_shard_gpu.py (run under torch.distributed.run): ONE linear system split by rows across the ranks
(scs_amd/shard.py), checked on rank 0 against the unsplit solve of libscsamd_linsys.so and, when built, the reference backend.
argv: backend n m col_nnz"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from scs_amd import capi, shard
from tests import probgen

backend, n, m, col_nnz = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
torch.cuda.set_device(local_rank)
lib = capi.load("libscsamd_linsys.so")
assert lib.scs_amd_set_device(local_rank) == 0
dist.init_process_group(backend=backend)
rank, world = dist.get_rank(), dist.get_world_size()
A = probgen.random_csc(m, n, col_nnz, seed=5)
dr = probgen.diag_r(n, m, z=m // 10)
rng = np.random.default_rng(3)
b = rng.uniform(-1, 1, n + m)
s = rng.uniform(-1, 1, n) * 0.1
S = shard.ShardedLinSys(A, dr, dist=dist, device="cuda", lib=lib)
x, y_loc = S.solve(b, s, tol=1e-11)
y = S.gather_y(y_loc)
x2, _ = S.solve(2 * b, None, tol=1e-11)   # a second solve on the same workspaces, cold start
out = dict(rank=rank, world=world, backend=dist.get_backend(), rows=[S.r0, S.r1], cg_iters=S.cg_iters, allreduce_calls=S.allreduce_calls)
if rank == 0:
    T = lib._scs_types
    prob = capi.Problem(A, np.zeros(m), np.zeros(n), dict(l=m))
    w = lib.scs_init_lin_sys_work(C.byref(prob.matA), None, dr.ctypes.data_as(T.fp))
    o = b.copy()
    assert lib.scs_solve_lin_sys(w, o.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp), 1e-11) == 0
    o2 = 2 * b
    assert lib.scs_solve_lin_sys(w, o2.ctypes.data_as(T.fp), None, 1e-11) == 0
    lib.scs_free_lin_sys_work(w)
    sc = np.abs(o).max()
    out.update(err_x=float(np.abs(x - o[:n]).max() / sc), err_y=float(np.abs(y - o[n:]).max() / sc),
               err_x2=float(np.abs(x2 - o2[:n]).max() / np.abs(o2).max()))
    from oracle import pyoracle
    if pyoracle.ref_available():
        ref = pyoracle.load_ref()
        wr = ref.scs_init_lin_sys_work(C.byref(prob.matA), None, dr.ctypes.data_as(T.fp))
        orf = b.copy()
        assert ref.scs_solve_lin_sys(wr, orf.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp), 1e-11) == 0
        ref.scs_free_lin_sys_work(wr)
        out["err_vs_reference"] = float(np.abs(np.concatenate([x, y]) - orf).max() / np.abs(orf).max())
    print("SHARD " + json.dumps(out), flush=True)
S.close()
dist.barrier()
dist.destroy_process_group()
