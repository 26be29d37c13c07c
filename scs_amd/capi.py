"""ctypes view of the C ABI declared in include/scs_amd.h.

The structs mirror reference include/scs.h:47-244 field for field, so the same
Python objects can be handed to libscsamd.so (this repo, HIP) and -- from tests
only -- to a build of the reference.  Nothing here computes anything: it is the
binding a scs-python maintainer would write.

The product library is loaded from scs_amd/lib/ (built in-tree by
``__graft_entry__.build()``); a missing library raises, there is no fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

scs_int = C.c_int


def make_types(ftype, scs_int=C.c_int):
    """Build the struct family for scs_float = ftype (c_double or c_float) and scs_int (c_int, or c_longlong
    for the -DDLONG library)."""
    fp = C.POINTER(ftype)
    ip = C.POINTER(scs_int)

    class ScsMatrix(C.Structure):
        _fields_ = [("x", fp), ("i", ip), ("p", ip), ("m", scs_int), ("n", scs_int)]

    class ScsSettings(C.Structure):
        _fields_ = [
            ("normalize", scs_int), ("scale", ftype), ("adaptive_scale", scs_int),
            ("rho_x", ftype), ("max_iters", scs_int), ("eps_abs", ftype),
            ("eps_rel", ftype), ("eps_infeas", ftype), ("alpha", ftype),
            ("time_limit_secs", ftype), ("verbose", scs_int), ("warm_start", scs_int),
            ("acceleration_lookback", scs_int), ("acceleration_interval", scs_int),
            ("acceleration_type_1", scs_int), ("acceleration_regularization", ftype),
            ("acceleration_relaxation", ftype),
            ("write_data_filename", C.c_char_p), ("log_csv_filename", C.c_char_p),
        ]

    class ScsData(C.Structure):
        _fields_ = [("m", scs_int), ("n", scs_int), ("A", C.POINTER(ScsMatrix)),
                    ("P", C.POINTER(ScsMatrix)), ("b", fp), ("c", fp)]

    class ScsCone(C.Structure):
        _fields_ = [
            ("z", scs_int), ("l", scs_int), ("bu", fp), ("bl", fp), ("bsize", scs_int),
            ("q", ip), ("qsize", scs_int), ("s", ip), ("ssize", scs_int),
            ("cs", ip), ("cssize", scs_int), ("ep", scs_int), ("ed", scs_int),
            ("p", fp), ("psize", scs_int),
        ]

    class ScsSolution(C.Structure):
        _fields_ = [("x", fp), ("y", fp), ("s", fp)]

    class AaStats(C.Structure):
        _fields_ = [
            ("iter", scs_int), ("n_accept", scs_int), ("n_reject_lapack", scs_int),
            ("n_reject_rank0", scs_int), ("n_reject_nonfinite", scs_int),
            ("n_reject_weight_cap", scs_int), ("n_safeguard_reject", scs_int),
            ("last_rank", scs_int), ("last_aa_norm", ftype), ("last_regularization", ftype),
        ]

    class ScsInfo(C.Structure):
        _fields_ = [
            ("iter", scs_int), ("status", C.c_char * 128), ("lin_sys_solver", C.c_char * 128),
            ("status_val", scs_int), ("scale_updates", scs_int), ("pobj", ftype),
            ("dobj", ftype), ("res_pri", ftype), ("res_dual", ftype), ("gap", ftype),
            ("res_infeas", ftype), ("res_unbdd_a", ftype), ("res_unbdd_p", ftype),
            ("setup_time", ftype), ("solve_time", ftype), ("scale", ftype),
            ("comp_slack", ftype), ("rejected_accel_steps", scs_int),
            ("accepted_accel_steps", scs_int), ("aa_stats", AaStats),
            ("lin_sys_time", ftype), ("cone_time", ftype), ("accel_time", ftype),
        ]

    class ScsAmdStats(C.Structure):
        _fields_ = [
            ("cg_iters", C.c_longlong), ("lin_sys_solves", C.c_longlong),
            ("mat_vecs", C.c_longlong), ("spmv_launches", C.c_longlong),
            ("spmv_ms", C.c_double), ("cg_ms", C.c_double), ("cone_ms", C.c_double),
            ("cone_projs", C.c_longlong), ("nnz", C.c_longlong), ("spmv_bytes", C.c_longlong),
            ("psd_unconverged", C.c_longlong),
        ]

    ns = type("ScsTypes", (), {})
    ns.ftype, ns.fp, ns.ip, ns.scs_int = ftype, fp, ip, scs_int
    ns.np_float = np.float64 if ftype is C.c_double else np.float32
    ns.np_int = np.int32 if scs_int is C.c_int else np.int64
    ns.ScsMatrix, ns.ScsSettings, ns.ScsData, ns.ScsCone = ScsMatrix, ScsSettings, ScsData, ScsCone
    ns.ScsSolution, ns.AaStats, ns.ScsInfo, ns.ScsAmdStats = ScsSolution, AaStats, ScsInfo, ScsAmdStats
    return ns


T64 = make_types(C.c_double)
T32 = make_types(C.c_float)
T64L = make_types(C.c_double, C.c_longlong)  # libscsamd_dlong.so


def bind_api(lib, T, full=True, linsys=True, cones=True, stats=True):
    """Attach argtypes/restypes for every entry point the library exports."""
    fp = T.fp
    scs_int = T.scs_int
    if full:
        lib.scs_init.restype = C.c_void_p
        lib.scs_init.argtypes = [C.POINTER(T.ScsData), C.POINTER(T.ScsCone), C.POINTER(T.ScsSettings)]
        lib.scs_update.restype = scs_int
        lib.scs_update.argtypes = [C.c_void_p, fp, fp]
        lib.scs_solve.restype = scs_int
        lib.scs_solve.argtypes = [C.c_void_p, C.POINTER(T.ScsSolution), C.POINTER(T.ScsInfo), scs_int]
        lib.scs_finish.restype = None
        lib.scs_finish.argtypes = [C.c_void_p]
        lib.scs.restype = scs_int
        lib.scs.argtypes = [C.POINTER(T.ScsData), C.POINTER(T.ScsCone), C.POINTER(T.ScsSettings),
                            C.POINTER(T.ScsSolution), C.POINTER(T.ScsInfo)]
        lib.scs_set_default_settings.restype = None
        lib.scs_set_default_settings.argtypes = [C.POINTER(T.ScsSettings)]
        lib.scs_version.restype = C.c_char_p
        lib.scs_version.argtypes = []
    if linsys:
        lib.scs_init_lin_sys_work.restype = C.c_void_p
        lib.scs_init_lin_sys_work.argtypes = [C.POINTER(T.ScsMatrix), C.POINTER(T.ScsMatrix), fp]
        lib.scs_solve_lin_sys.restype = scs_int
        lib.scs_solve_lin_sys.argtypes = [C.c_void_p, fp, fp, T.ftype]
        lib.scs_update_lin_sys_diag_r.restype = scs_int
        lib.scs_update_lin_sys_diag_r.argtypes = [C.c_void_p, fp]
        lib.scs_free_lin_sys_work.restype = None
        lib.scs_free_lin_sys_work.argtypes = [C.c_void_p]
        lib.scs_get_lin_sys_method.restype = C.c_char_p
        lib.scs_get_lin_sys_method.argtypes = []
    if cones:
        lib.scs_amd_cone_init.restype = C.c_void_p
        lib.scs_amd_cone_init.argtypes = [C.POINTER(T.ScsCone), scs_int, fp]
        lib.scs_amd_cone_proj_dual.restype = scs_int
        lib.scs_amd_cone_proj_dual.argtypes = [C.c_void_p, fp, fp]
        lib.scs_amd_cone_finish.restype = None
        lib.scs_amd_cone_finish.argtypes = [C.c_void_p]
    if stats:
        lib.scs_amd_linsys_get_stats.restype = None
        lib.scs_amd_linsys_get_stats.argtypes = [C.c_void_p, C.POINTER(T.ScsAmdStats)]
        lib.scs_amd_linsys_set_profiling.restype = None
        lib.scs_amd_linsys_set_profiling.argtypes = [C.c_void_p, scs_int]
        for nm in ("scs_amd_linsys_mat_vec_dev", "scs_amd_linsys_mul_a_dev", "scs_amd_linsys_mul_at_dev"):
            fn = getattr(lib, nm)
            fn.restype = scs_int
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]  # device pointers
        lib.scs_amd_linsys_sync.restype = scs_int
        lib.scs_amd_linsys_sync.argtypes = [C.c_void_p]
        lib.scs_amd_set_option.restype = scs_int
        lib.scs_amd_set_option.argtypes = [C.c_char_p, C.c_char_p]
        lib.scs_amd_get_option.restype = C.c_char_p
        lib.scs_amd_get_option.argtypes = [C.c_char_p]
        lib.scs_amd_list_options.restype = scs_int
        lib.scs_amd_list_options.argtypes = [C.c_char_p, scs_int]
        lib.scs_amd_device_free_bytes.restype = C.c_longlong
        lib.scs_amd_device_free_bytes.argtypes = []
        lib.scs_amd_test_fail_at.restype = C.c_longlong
        lib.scs_amd_test_fail_at.argtypes = [C.c_longlong]
        lib.scs_amd_device_count.restype = scs_int
        lib.scs_amd_device_count.argtypes = []
        lib.scs_amd_set_device.restype = scs_int
        lib.scs_amd_set_device.argtypes = [scs_int]
        if full:
            lib.scs_amd_get_stats.restype = None
            lib.scs_amd_get_stats.argtypes = [C.c_void_p, C.POINTER(T.ScsAmdStats)]
            lib.scs_amd_set_profiling.restype = None
            lib.scs_amd_set_profiling.argtypes = [C.c_void_p, scs_int]
            lib.scs_amd_solve_begin.restype = scs_int
            lib.scs_amd_solve_begin.argtypes = [C.c_void_p, C.POINTER(T.ScsSolution), scs_int]
            lib.scs_amd_solve_steps.restype = scs_int
            lib.scs_amd_solve_steps.argtypes = [C.c_void_p, scs_int]
            lib.scs_amd_solve_converged.restype = scs_int
            lib.scs_amd_solve_converged.argtypes = [C.c_void_p]
            lib.scs_amd_solve_end.restype = scs_int
            lib.scs_amd_solve_end.argtypes = [C.c_void_p, C.POINTER(T.ScsSolution), C.POINTER(T.ScsInfo)]
            lib.scs_amd_set_cg_tol_override.restype = None
            lib.scs_amd_set_cg_tol_override.argtypes = [C.c_void_p, C.c_double]
            # one linear system split by rows across GPUs, native form (scs_amd/csrc/shard_native.cpp)
            lib.scs_amd_shard_unique_id.restype = scs_int
            lib.scs_amd_shard_unique_id.argtypes = [C.c_char_p]
            lib.scs_amd_shard_init_rccl.restype = C.c_void_p
            lib.scs_amd_shard_init_rccl.argtypes = [C.POINTER(T.ScsMatrix), fp, scs_int, scs_int, C.c_char_p]
            lib.scs_amd_shard_group_create.restype = C.c_void_p
            lib.scs_amd_shard_group_create.argtypes = [scs_int]
            lib.scs_amd_shard_group_free.restype = None
            lib.scs_amd_shard_group_free.argtypes = [C.c_void_p]
            lib.scs_amd_shard_init_threads.restype = C.c_void_p
            lib.scs_amd_shard_init_threads.argtypes = [C.POINTER(T.ScsMatrix), fp, C.c_void_p, scs_int]
            lib.scs_amd_shard_solve.restype = scs_int
            lib.scs_amd_shard_solve.argtypes = [C.c_void_p, fp, fp, T.ftype]
            lib.scs_amd_shard_update_diag_r.restype = scs_int
            lib.scs_amd_shard_update_diag_r.argtypes = [C.c_void_p, fp]
            lib.scs_amd_shard_get_stats.restype = None
            lib.scs_amd_shard_get_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
            lib.scs_amd_shard_set_profiling.restype = None
            lib.scs_amd_shard_set_profiling.argtypes = [C.c_void_p, scs_int]
            lib.scs_amd_shard_free.restype = None
            lib.scs_amd_shard_free.argtypes = [C.c_void_p]
            lib.scs_amd_plan_reorder.restype = scs_int
            lib.scs_amd_plan_reorder.argtypes = [C.POINTER(T.ScsMatrix), C.POINTER(T.ScsCone), T.ip, T.ip, C.POINTER(C.c_double)]
            lib.scs_amd_get_reorder_info.restype = None
            lib.scs_amd_get_reorder_info.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
            lib.scs_amd_get_layout_info.restype = None
            lib.scs_amd_get_layout_info.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
            lib.scs_amd_get_spmv_kernel_name.restype = scs_int
            lib.scs_amd_get_spmv_kernel_name.argtypes = [C.c_void_p, scs_int, C.c_char_p, scs_int]
            lib.scs_amd_set_residuals_every_iter.restype = None
            lib.scs_amd_set_residuals_every_iter.argtypes = [C.c_void_p, scs_int]
    lib._scs_types = T
    return lib


_cache = {}


def lib_path(name):
    return os.path.join(LIB_DIR, name)


def load(name="libscsamd.so"):
    """Load a product library from scs_amd/lib/.  Raises if it was not built."""
    if name in _cache:
        return _cache[name]
    path = lib_path(name)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback in the product path.")
    lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    T = T32 if name.endswith("_f32.so") else (T64L if name.endswith("_dlong.so") else T64)
    only_linsys = "linsys" in name
    bind_api(lib, T, full=not only_linsys, linsys=True, cones=not only_linsys, stats=True)
    _cache[name] = lib
    return lib


def set_option(key, value, libs=None):
    """scs_amd_set_option on every product library loaded so far (each shared object keeps its own table): process-wide,
    in force for workspaces created afterwards.  value None = back to the default.  Raises on a key the table does not hold."""
    for name, lib in list(_cache.items()) if libs is None else [(None, l) for l in libs]:
        rc = lib.scs_amd_set_option(key.encode(), None if value is None else str(value).encode())
        if rc != 0:
            raise KeyError(f"scs_amd: unknown option {key!r}")


def list_options(lib=None):
    """The option table of scs_amd/csrc/options.h as a list of dicts (key, cls, numerics, values, doc)."""
    lib = lib or load("libscsamd.so")
    n = lib.scs_amd_list_options(None, 0)
    buf = C.create_string_buffer(n + 1)
    lib.scs_amd_list_options(buf, n + 1)
    rows = []
    for line in buf.value.decode().splitlines():
        k, cls, num, vals, doc = line.split("\t")
        rows.append(dict(key=k, cls=cls, numerics=int(num), values=vals, doc=doc))
    return rows


# ---- numpy <-> struct helpers ---------------------------------------------------

class Problem:
    """Owns the numpy arrays behind one (ScsData, ScsCone) pair so ctypes
    pointers stay valid.  A: scipy.sparse CSC (m x n); P: upper-tri CSC or None."""

    def __init__(self, A, b, c, cone, P=None, T=T64):
        import scipy.sparse as sp
        self.T = T
        f = T.np_float
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.m, self.n = A.shape
        self.Ax = np.ascontiguousarray(A.data, dtype=f)
        self.Ai = np.ascontiguousarray(A.indices, dtype=T.np_int)
        self.Ap = np.ascontiguousarray(A.indptr, dtype=T.np_int)
        self.b = np.ascontiguousarray(b, dtype=f)
        self.c = np.ascontiguousarray(c, dtype=f)
        self.cone = dict(cone)
        self.matA = T.ScsMatrix(self.Ax.ctypes.data_as(T.fp), self.Ai.ctypes.data_as(T.ip),
                                self.Ap.ctypes.data_as(T.ip), self.m, self.n)
        self.matP = None
        if P is not None:
            P = sp.csc_matrix(sp.triu(P))
            P.sort_indices()
            self.Px = np.ascontiguousarray(P.data, dtype=f)
            self.Pi = np.ascontiguousarray(P.indices, dtype=T.np_int)
            self.Pp = np.ascontiguousarray(P.indptr, dtype=T.np_int)
            self.matP = T.ScsMatrix(self.Px.ctypes.data_as(T.fp), self.Pi.ctypes.data_as(T.ip),
                                    self.Pp.ctypes.data_as(T.ip), self.n, self.n)
        self.data = T.ScsData(self.m, self.n, C.pointer(self.matA),
                              C.pointer(self.matP) if self.matP is not None else None,
                              self.b.ctypes.data_as(T.fp), self.c.ctypes.data_as(T.fp))
        self.k = make_cone(self.cone, T, keep=self)

    def sparse(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.Ax, self.Ai, self.Ap), shape=(self.m, self.n))


def make_cone(cone, T=T64, keep=None):
    """dict(z=, l=, bu=, bl=, q=[...], s=[...]) -> ScsCone (arrays kept alive on `keep`)."""
    f = T.np_float
    k = T.ScsCone()
    holder = keep if keep is not None else k
    k.z = int(cone.get("z", 0))
    k.l = int(cone.get("l", 0))
    bu = np.ascontiguousarray(cone.get("bu", []), dtype=f)
    bl = np.ascontiguousarray(cone.get("bl", []), dtype=f)
    assert len(bu) == len(bl)
    k.bsize = int(cone.get("bsize", len(bu) + 1 if len(bu) else 0))
    q = np.ascontiguousarray(cone.get("q", []), dtype=T.np_int)
    s = np.ascontiguousarray(cone.get("s", []), dtype=T.np_int)
    pw = np.ascontiguousarray(cone.get("p", []), dtype=f)
    cs = np.ascontiguousarray(cone.get("cs", []), dtype=T.np_int)
    holder._cone_arrays = (bu, bl, q, s, pw, cs)
    k.bu = bu.ctypes.data_as(T.fp) if len(bu) else None
    k.bl = bl.ctypes.data_as(T.fp) if len(bl) else None
    k.q = q.ctypes.data_as(T.ip) if len(q) else None
    k.qsize = len(q)
    k.s = s.ctypes.data_as(T.ip) if len(s) else None
    k.ssize = len(s)
    k.cs = cs.ctypes.data_as(T.ip) if len(cs) else None
    k.cssize = len(cs)
    k.ep, k.ed = int(cone.get("ep", 0)), int(cone.get("ed", 0))
    k.p = pw.ctypes.data_as(T.fp) if len(pw) else None
    k.psize = len(pw)
    return k


def cone_rows(cone):
    """Total number of rows a cone dict covers."""
    q = list(cone.get("q", []))
    s = list(cone.get("s", []))
    bs = cone.get("bsize", len(cone.get("bu", [])) + 1 if len(cone.get("bu", [])) else 0)
    return int(cone.get("z", 0) + cone.get("l", 0) + bs + sum(q) + sum(v * (v + 1) // 2 for v in s) +
               sum(int(v) ** 2 for v in cone.get("cs", [])) +
               3 * (cone.get("ep", 0) + cone.get("ed", 0) + len(cone.get("p", []))))


def default_settings(lib, **over):
    T = lib._scs_types
    st = T.ScsSettings()
    lib.scs_set_default_settings(C.byref(st))
    for k, v in over.items():
        if not hasattr(st, k):
            raise AttributeError(k)
        setattr(st, k, v)
    return st


INFO_FIELDS = ["iter", "status_val", "scale_updates", "pobj", "dobj", "res_pri", "res_dual", "gap",
               "res_infeas", "res_unbdd_a", "res_unbdd_p", "setup_time", "solve_time", "scale",
               "comp_slack", "rejected_accel_steps", "accepted_accel_steps", "lin_sys_time",
               "cone_time", "accel_time"]


def info_dict(info):
    d = {k: getattr(info, k) for k in INFO_FIELDS}
    d["status"] = info.status.decode()
    d["lin_sys_solver"] = info.lin_sys_solver.decode()
    return d


def solve(lib, prob, settings=None, warm=None, want_stats=False, profiling=False, cg_tol_override=None, **over):
    """scs_init -> scs_solve -> scs_finish through the C ABI of `lib`.
    Returns dict(x, y, s, info[, stats])."""
    T = lib._scs_types
    st = settings if settings is not None else default_settings(lib, **over)
    f = T.np_float
    x = np.zeros(prob.n, dtype=f)
    y = np.zeros(prob.m, dtype=f)
    s = np.zeros(prob.m, dtype=f)
    if warm is not None:
        x[:], y[:], s[:] = warm
    sol = T.ScsSolution(x.ctypes.data_as(T.fp), y.ctypes.data_as(T.fp), s.ctypes.data_as(T.fp))
    info = T.ScsInfo()
    w = lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))
    if not w:
        raise RuntimeError("scs_init returned NULL")
    try:
        if profiling and hasattr(lib, "scs_amd_set_profiling"):
            lib.scs_amd_set_profiling(w, 1)
        if cg_tol_override is not None:
            lib.scs_amd_set_cg_tol_override(w, float(cg_tol_override))  # test hook (HIP library only)
        lib.scs_solve(w, C.byref(sol), C.byref(info), 1 if warm is not None else 0)
        out = dict(x=x, y=y, s=s, info=info_dict(info))
        if want_stats and hasattr(lib, "scs_amd_get_stats"):
            stt = T.ScsAmdStats()
            lib.scs_amd_get_stats(w, C.byref(stt))
            out["stats"] = {k: getattr(stt, k) for k, _ in T.ScsAmdStats._fields_}
    finally:
        lib.scs_finish(w)
    return out
