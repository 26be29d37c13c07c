"""Worker of tests/test_host_logic.py::test_row_sharded_pcg_over_two_gloo_ranks (CPU only, run under torch.distributed.run):
the collective / PCG logic of scs_amd/shard.py with the slab's operator pieces played by scipy (the GPU kernels are tested in
tests/test_shard_gpu.py): two ranks, one all-reduce of an n-vector per CG iteration over gloo."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from scs_amd import shard
from tests import probgen


class ScipySlabOps:
    """stand-in for HipSlabOps: the same three products on CPU tensors"""

    def __init__(self, Ar, rx_share, ry):
        self.A, self.rx, self.ry = Ar.tocsr(), rx_share, ry

    def mat_vec(self, x, y):
        v = x.numpy()
        y.copy_(torch.from_numpy(self.rx * v + self.A.T @ ((self.A @ v) / self.ry)))

    def mul_a(self, x, y):
        y.copy_(torch.from_numpy(self.A @ x.numpy()))

    def mul_at(self, yv, x):
        x.copy_(torch.from_numpy(self.A.T @ yv.numpy()))

    def close(self):
        pass


dist.init_process_group(backend="gloo")
n, m = 300, 801
A = probgen.random_csc(m, n, 6, seed=2)
dr = probgen.diag_r(n, m, z=m // 10)
rng = np.random.default_rng(1)
b = rng.uniform(-1, 1, n + m)
s = rng.uniform(-1, 1, n) * 0.1
S = shard.ShardedLinSys(A, dr, dist=dist, device="cpu", ops_factory=ScipySlabOps)
x, y_loc = S.solve(b, s, tol=1e-12)
y = S.gather_y(y_loc)
if dist.get_rank() == 0:
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    K = sp.bmat([[sp.diags(dr[:n]), A.T], [A, -sp.diags(dr[n:])]]).tocsc()  # the KKT system of linsys/cpu/indirect/private.c:284-324
    want = spla.spsolve(K, b)
    err = float(np.abs(np.concatenate([x, y]) - want).max() / np.abs(want).max())
    print("SHARDCPU " + json.dumps(dict(world=dist.get_world_size(), rows=[S.r0, S.r1], cg_iters=S.cg_iters, allreduce_calls=S.allreduce_calls, err=err)), flush=True)
S.close()
dist.barrier()
dist.destroy_process_group()
