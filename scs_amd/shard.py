"""ONE linear system split by rows of A across ranks ("intra-problem sharding"; the operator that is
split is reference linsys/cpu/indirect/private.c:106-119, the solve is :133-324).

Rank r holds the row slab A_r (rows [r0, r1) of A, all n columns) and the matching slab of R_y; every rank holds the
n-vectors.  The reduced operator is a sum over ranks,

    G x = R_x x + A' R_y^-1 A x = sum_r ( (R_x / N) x + A_r' R_r^-1 A_r x ),

so one PCG iteration is: every rank applies ITS term with the MI355X SpMV kernels (`scs_amd_linsys_mat_vec_dev`, device
pointers), one all-reduce of an n-vector (RCCL: `torch.distributed` backend "nccl"; gloo in the CPU-hosted tests) forms G p,
and the level-1 part of the iteration runs redundantly on every rank (so no second collective is needed for alpha / beta).
The back-substitution y = R_y^-1 (A x - r_y) is local to each slab.

This is the FUNCTIONAL form of the split (torch is plumbing here: device vectors, the collective, the O(n) vector
algebra); it is not the product path of the benchmark: at n = 1e6 the all-reduce moves 8 MB per CG iteration against
~170 us of compute, i.e. ~100 us on 7 x 153 GB/s xGMI links -- a split that cannot pay at these sizes.
It exists for problems that do not fit one GPU's 288 GB, and to have the data-path collective exercised.
"""
import ctypes as C

import numpy as np

from . import capi


def slab(m, world, rank):
    """rows [r0, r1) of rank `rank`: contiguous, sizes differing by at most one"""
    base, extra = divmod(m, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


class HipSlabOps:
    """the slab's operator pieces on the GPU: a B1 workspace of libscsamd_linsys.so driven through its device-pointer entries"""

    def __init__(self, Ar, rx_share, ry, lib=None, device=None):
        import torch
        self.torch = torch
        self.lib = lib or capi.load("libscsamd_linsys.so")
        self.T = T = self.lib._scs_types
        # the workspace must live on the GPU the caller's tensors live on: bind the library's device selection to torch's
        # (without this every slab landed on the library's default device 0 while x / y belonged to another GPU)
        self.device_index = torch.cuda.current_device() if device is None else torch.device(device).index
        if self.device_index is None:
            self.device_index = torch.cuda.current_device()
        if self.lib.scs_amd_set_device(self.device_index) != 0:
            raise RuntimeError(f"scs_amd_set_device({self.device_index}) failed")
        mr, n = Ar.shape
        self.prob = capi.Problem(Ar, np.zeros(mr), np.zeros(n), dict(l=mr), T=T)
        local = np.concatenate([rx_share, ry]).astype(T.np_float)
        self.w = self.lib.scs_init_lin_sys_work(C.byref(self.prob.matA), None, local.ctypes.data_as(T.fp))
        if not self.w:
            raise RuntimeError("scs_init_lin_sys_work failed on the row slab")

    def _call(self, fn, src, dst):
        for t in (src, dst):  # device pointers are handed to kernels of THIS workspace's device, in the library's precision
            if not t.is_cuda or t.device.index != self.device_index:
                raise ValueError(f"tensor on {t.device}, workspace on cuda:{self.device_index}")
            if t.element_size() != np.dtype(self.T.np_float).itemsize or not t.is_contiguous():
                raise ValueError("tensor dtype / layout does not match the library's scs_float")
        self.torch.cuda.synchronize()  # torch's stream and the workspace's stream are different streams: functional form
        if fn(self.w, src.data_ptr(), dst.data_ptr()) != 0:
            raise RuntimeError("device operator call failed")
        if self.lib.scs_amd_linsys_sync(self.w) != 0:
            raise RuntimeError("stream synchronisation failed")

    def mat_vec(self, x, y):
        self._call(self.lib.scs_amd_linsys_mat_vec_dev, x, y)

    def mul_a(self, x, y):
        self._call(self.lib.scs_amd_linsys_mul_a_dev, x, y)

    def mul_at(self, y, x):
        self._call(self.lib.scs_amd_linsys_mul_at_dev, y, x)

    def close(self):
        if self.w:
            self.lib.scs_free_lin_sys_work(self.w)
            self.w = None


class ShardedLinSys:
    """A: scipy CSC (m x n) -- every rank passes the same matrix (or at least its own rows); diag_r = [R_x (n); R_y (m)].
    `ops_factory(Ar, rx_share, ry)` builds the slab's operator pieces (default: HipSlabOps, the GPU; the CPU-hosted gloo test
    of the collective / PCG logic injects a scipy stand-in)."""

    def __init__(self, A, diag_r, dist=None, device="cuda", lib=None, ops_factory=None, dtype=np.float64):
        import scipy.sparse as sp
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.world = 1 if dist is None else dist.get_world_size()
        self.rank = 0 if dist is None else dist.get_rank()
        self.m, self.n = A.shape
        self.r0, self.r1 = slab(self.m, self.world, self.rank)
        Ar = sp.csc_matrix(sp.csr_matrix(A)[self.r0:self.r1, :])
        f = dtype
        diag_r = np.asarray(diag_r, dtype=f)
        self.rx = diag_r[:self.n].copy()
        self.ry = diag_r[self.n + self.r0:self.n + self.r1].copy()
        if ops_factory is None:
            want = (lib or capi.load("libscsamd_linsys.so"))._scs_types.np_float
            if np.dtype(f) != np.dtype(want):
                raise ValueError(f"dtype {np.dtype(f)} does not match the library's scs_float ({np.dtype(want)})")
        self.ops = (ops_factory or (lambda a, b, c: HipSlabOps(a, b, c, lib, device=device)))(Ar, self.rx / self.world, self.ry)
        td = torch.float64 if f is np.float64 else torch.float32
        self.td = td
        # Jacobi preconditioner (private.c:50-82): diag(G) = R_x + sum_r diag(A_r' R_r^-1 A_r)
        Asq = Ar.multiply(Ar)
        d = torch.tensor(np.asarray(Asq.T @ (1.0 / self.ry)).ravel(), dtype=td, device=device)
        self._allreduce(d)
        self.M = 1.0 / (torch.tensor(self.rx, dtype=td, device=device) + d)
        self.ry_d = torch.tensor(self.ry, dtype=td, device=device)
        self.allreduce_calls = 0
        self.cg_iters = 0

    # ---- plumbing -----------------------------------------------------------------------------------------------
    def _allreduce(self, t):
        """in-place sum over ranks of a device tensor (RCCL when the group's backend is nccl; gloo goes through the host)"""
        if self.dist is None:
            return t
        if self.dist.get_backend() == "nccl":
            self.dist.all_reduce(t)
        else:
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        return t

    def G(self, x):
        """G x, the same vector on every rank"""
        y = self.torch.empty_like(x)
        self.ops.mat_vec(x, y)
        self.allreduce_calls += 1
        return self._allreduce(y)

    # ---- scs_solve_lin_sys (private.c:284-324) --------------------------------------------------------------------
    def solve(self, b, s=None, tol=1e-9):
        """b = [r_x (n); r_y (m)] (host, the same on every rank), s = warm start (n) or None.
        Returns (x (n), y slab (r1 - r0)) as numpy arrays: x identical on every rank, y the rank's rows."""
        torch, td, dev, n = self.torch, self.td, self.device, self.n
        b = np.asarray(b)
        if np.abs(b).max() <= 1e-12:  # private.c:296-299
            return np.zeros(n), np.zeros(self.r1 - self.r0)
        bx = torch.tensor(b[:n], dtype=td, device=dev)
        by = torch.tensor(b[n + self.r0:n + self.r1], dtype=td, device=dev)
        t = torch.empty(n, dtype=td, device=dev)
        self.ops.mul_at(by / self.ry_d, t)  # A_r' R_r^-1 r_y
        bx = bx + self._allreduce(t)
        if s is not None:
            x = torch.tensor(np.asarray(s), dtype=td, device=dev)
            r = bx - self.G(x)
        else:
            x = torch.zeros(n, dtype=td, device=dev)
            r = bx.clone()
        z = self.M * r
        ztr = torch.dot(z, r)
        its = 0
        if float(r.abs().max()) >= max(tol, 1e-12):  # private.c:163
            p = z.clone()
            for its in range(1, 10 * n + 1):  # private.c:174-217
                Gp = self.G(p)
                alpha = ztr / torch.dot(p, Gp)
                x += alpha * p
                r -= alpha * Gp
                if float(r.abs().max()) < tol:
                    break
                z = self.M * r
                ztr_prev, ztr = ztr, torch.dot(z, r)
                if float(ztr_prev) == 0.0:
                    its -= 1
                    break
                p = z + (ztr / ztr_prev) * p
        self.cg_iters += its
        ax = torch.empty(self.r1 - self.r0, dtype=td, device=dev)
        self.ops.mul_a(x, ax)
        y = (ax - by) / self.ry_d  # private.c:313-317
        return x.cpu().numpy(), y.cpu().numpy()

    def gather_y(self, y_local):
        """every rank's slab of y in row order (host)"""
        if self.dist is None:
            return y_local
        parts = [None] * self.world
        self.dist.all_gather_object(parts, y_local)
        return np.concatenate(parts)

    def close(self):
        self.ops.close()


class NativeShardedLinSys:
    """The same split in its NATIVE form (scs_amd/csrc/shard_native.cpp, include/scs_amd.h `scs_amd_shard_*`): the PCG loop is the
    device-controlled loop of the unsplit solver, the all-reduce of G p is enqueued on the solver's own stream through RCCL's C API
    (or the in-process test double when `group` is given), nothing is read back per iteration.  One instance per rank; every call
    is collective.  A: scipy sparse (m x n), the full matrix (this rank keeps its slab); diag_r = [R_x (n); R_y (m)]."""

    def __init__(self, A, diag_r, world=1, rank=0, unique_id=None, group=None, lib=None):
        import scipy.sparse as sp
        self.lib = lib or capi.load("libscsamd.so")
        self.T = T = self.lib._scs_types
        self.m, self.n = A.shape
        self.world, self.rank = world, rank
        self.r0, self.r1 = slab(self.m, world, rank)
        Ar = sp.csc_matrix(sp.csr_matrix(A)[self.r0:self.r1, :])
        mr = self.r1 - self.r0
        self.prob = capi.Problem(Ar, np.zeros(mr), np.zeros(self.n), dict(l=mr), T=T)
        diag_r = np.asarray(diag_r, dtype=T.np_float)
        local = np.concatenate([diag_r[:self.n] / world, diag_r[self.n + self.r0:self.n + self.r1]]).astype(T.np_float)
        if group is not None:
            self.h = self.lib.scs_amd_shard_init_threads(C.byref(self.prob.matA), local.ctypes.data_as(T.fp), group, rank)
        else:
            if unique_id is None:
                if world != 1:
                    raise ValueError("unique_id of rank 0 (NativeShardedLinSys.unique_id()) must be handed to every rank")
                unique_id = self.unique_id(self.lib)
            self.h = self.lib.scs_amd_shard_init_rccl(C.byref(self.prob.matA), local.ctypes.data_as(T.fp), world, rank, unique_id)
        if not self.h:
            raise RuntimeError("scs_amd_shard_init failed")

    @staticmethod
    def unique_id(lib=None):
        lib = lib or capi.load("libscsamd.so")
        buf = C.create_string_buffer(128)
        if lib.scs_amd_shard_unique_id(buf) != 0:
            raise RuntimeError("RCCL not available (scs_amd_shard_unique_id)")
        return buf.raw

    def solve(self, rhs, s=None, tol=1e-9):
        """rhs = [r_x (n); r_y (m)] (the full right-hand side; this rank uses r_x and its slab of r_y).
        Returns (x (n), y slab (m_r))."""
        T = self.T
        b = np.concatenate([rhs[:self.n], rhs[self.n + self.r0:self.n + self.r1]]).astype(T.np_float)
        sp_ = None if s is None else np.ascontiguousarray(s, dtype=T.np_float)
        rc = self.lib.scs_amd_shard_solve(self.h, b.ctypes.data_as(T.fp), None if sp_ is None else sp_.ctypes.data_as(T.fp), tol)
        if rc != 0:
            raise RuntimeError("scs_amd_shard_solve failed")
        return b[:self.n].copy(), b[self.n:].copy()

    def profiling(self, on):
        self.lib.scs_amd_shard_set_profiling(self.h, 1 if on else 0)

    def stats(self):
        out = (C.c_double * 5)()
        self.lib.scs_amd_shard_get_stats(self.h, out)
        return dict(cg_iters=int(out[0]), allreduces=int(out[1]), allreduces_timed=int(out[2]), allreduce_mean_us=out[3], solves=int(out[4]))

    def close(self):
        if self.h:
            self.lib.scs_amd_shard_free(self.h)
            self.h = None
