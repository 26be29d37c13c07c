"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes loaders for (a) the real reference built by oracle/Makefile into
oracle/_ref/ and (b) this repo's plain-C restatement oracle/liboracle.so.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

from scs_amd import capi  # struct definitions only (no compute)

_cache = {}


def ref_available(name="libscsindir_ref.so"):
    return os.path.exists(os.path.join(REF_DIR, name))


def load_ref(name="libscsindir_ref.so"):
    """The reference's own CPU indirect library (fp64 / _omp / _f32 flavours)."""
    if name in _cache:
        return _cache[name]
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run `make -C oracle ref` where /root/reference exists")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    T = capi.T32 if name.endswith("_f32.so") else capi.T64
    capi.bind_api(lib, T, full=True, linsys=True, cones=False, stats=False)
    # internal (but exported) symbols of the reference used as kernel-level oracles
    fp, ip = T.fp, T.ip
    lib._scs_init_cone.restype = C.c_void_p
    lib._scs_init_cone.argtypes = [C.POINTER(T.ScsCone), capi.scs_int]
    lib._scs_proj_dual_cone.restype = capi.scs_int
    lib._scs_proj_dual_cone.argtypes = [fp, C.c_void_p, C.c_void_p, fp]
    lib._scs_finish_cone.restype = None
    lib._scs_finish_cone.argtypes = [C.c_void_p]
    lib._scs_accum_by_atrans.restype = None
    lib._scs_accum_by_atrans.argtypes = [C.POINTER(T.ScsMatrix), fp, fp]
    lib._scs_accum_by_a.restype = None
    lib._scs_accum_by_a.argtypes = [C.POINTER(T.ScsMatrix), fp, fp]
    _cache[name] = lib
    return lib


def build_restatement():
    subprocess.check_call(["make", "-C", _HERE, "restate"], stdout=subprocess.DEVNULL)


def load_restatement():
    """This repo's plain-C restatement of the hot path (oracle/scs_oracle.c)."""
    if "restate" in _cache:
        return _cache["restate"]
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build_restatement()
    lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    _cache["restate"] = lib
    return lib


# ---- typed wrappers around liboracle.so (the plain-C restatement) -----------------
import numpy as _np


class OrSettings(C.Structure):
    _fields_ = [("normalize", C.c_int), ("adaptive_scale", C.c_int), ("max_iters", C.c_int),
                ("scale", C.c_double), ("rho_x", C.c_double), ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("eps_infeas", C.c_double), ("alpha", C.c_double), ("cg_tol_override", C.c_double)]


class OrInfo(C.Structure):
    _fields_ = [("iter", C.c_int), ("status_val", C.c_int), ("scale_updates", C.c_int),
                ("pobj", C.c_double), ("dobj", C.c_double), ("res_pri", C.c_double), ("res_dual", C.c_double),
                ("gap", C.c_double), ("scale", C.c_double), ("cg_its", C.c_double)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None and len(a) else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None and len(a) else None


def restatement():
    lib = load_restatement()
    if getattr(lib, "_bound", False):
        return lib
    lib.or_linsys_init.restype = C.c_void_p
    lib.or_linsys_init.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp, _dp]
    lib.or_linsys_solve.restype = C.c_int
    lib.or_linsys_solve.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
    lib.or_linsys_update_diag_r.argtypes = [C.c_void_p, _dp]
    lib.or_linsys_tot_cg_its.restype = C.c_long
    lib.or_linsys_tot_cg_its.argtypes = [C.c_void_p]
    lib.or_linsys_free.argtypes = [C.c_void_p]
    lib.or_cone_init.restype = C.c_void_p
    lib.or_cone_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, _ip, C.c_int, _ip, _dp]
    lib.or_cone_set_extra.argtypes = [C.c_void_p, C.c_int, _ip, C.c_int, C.c_int, C.c_int, _dp]
    lib.or_cone_proj_dual.argtypes = [C.c_void_p, _dp, _dp]
    lib.or_cone_free.argtypes = [C.c_void_p]
    lib.or_accum_by_atrans.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
    lib.or_accum_by_a.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
    lib.or_solve.restype = C.c_int
    lib.or_solve.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp,
                             C.c_int, _ip, C.c_int, _ip, C.POINTER(OrSettings), _dp, _dp, _dp, C.POINTER(OrInfo)]
    lib.or_solve_ext.restype = C.c_int
    lib.or_solve_ext.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                 C.c_int, _ip, C.c_int, _ip, C.c_int, _ip, C.c_int, C.c_int, C.c_int, _dp,
                                 C.POINTER(OrSettings), _dp, _dp, _dp, C.POINTER(OrInfo)]
    lib._bound = True
    return lib


def _cone_args(cone):
    bu = _np.ascontiguousarray(cone.get("bu", []), dtype=_np.float64)
    bl = _np.ascontiguousarray(cone.get("bl", []), dtype=_np.float64)
    q = _np.ascontiguousarray(cone.get("q", []), dtype=_np.int32)
    s = _np.ascontiguousarray(cone.get("s", []), dtype=_np.int32)
    bsize = int(cone.get("bsize", len(bu) + 1 if len(bu) else 0))
    return int(cone.get("z", 0)), int(cone.get("l", 0)), bsize, bl, bu, q, s


def oracle_proj_dual_cone(cone, x, r_y=None, D=None):
    lib = restatement()
    z, l, bsize, bl, bu, q, s = _cone_args(cone)
    m = len(x)
    Dd = None if D is None else _np.ascontiguousarray(D, dtype=_np.float64)
    c = lib.or_cone_init(m, z, l, bsize, _d(bl), _d(bu), len(q), _i(q), len(s), _i(s), _d(Dd))
    cs = _np.ascontiguousarray(cone.get("cs", []), dtype=_np.int32)
    pw = _np.ascontiguousarray(cone.get("p", []), dtype=_np.float64)
    ep, ed = int(cone.get("ep", 0)), int(cone.get("ed", 0))
    if len(cs) or ep or ed or len(pw):
        lib.or_cone_set_extra(c, len(cs), _i(cs), ep, ed, len(pw), _d(pw))
    out = _np.array(x, dtype=_np.float64)
    r = None if r_y is None else _np.ascontiguousarray(r_y, dtype=_np.float64)
    lib.or_cone_proj_dual(c, _d(out), _d(r))
    lib.or_cone_free(c)
    return out


class OracleLinSys:
    def __init__(self, m, n, Ap, Ai, Ax, diag_r):
        self.lib = restatement()
        self.m, self.n = m, n
        self.Ap = _np.ascontiguousarray(Ap, dtype=_np.int32)
        self.Ai = _np.ascontiguousarray(Ai, dtype=_np.int32)
        self.Ax = _np.ascontiguousarray(Ax, dtype=_np.float64)
        self.dr = _np.ascontiguousarray(diag_r, dtype=_np.float64)
        self.w = self.lib.or_linsys_init(m, n, _i(self.Ap), _i(self.Ai), _d(self.Ax), _d(self.dr))

    def solve(self, b, s, tol):
        out = _np.array(b, dtype=_np.float64)
        ss = None if s is None or len(s) == 0 else _np.ascontiguousarray(s, dtype=_np.float64)
        before = self.lib.or_linsys_tot_cg_its(self.w)
        self.lib.or_linsys_solve(self.w, _d(out), _d(ss), float(tol))
        return out, self.lib.or_linsys_tot_cg_its(self.w) - before

    def update(self, diag_r):
        self.dr = _np.ascontiguousarray(diag_r, dtype=_np.float64)
        self.lib.or_linsys_update_diag_r(self.w, _d(self.dr))

    def close(self):
        if self.w:
            self.lib.or_linsys_free(self.w)
            self.w = None


def oracle_solve(prob, cg_tol_override=0.0, **over):
    """Whole ADMM solve by the restatement (AA off).  prob: scs_amd.capi.Problem (P must be None)."""
    lib = restatement()
    assert prob.matP is None
    st = OrSettings(normalize=1, adaptive_scale=1, max_iters=100000, scale=0.1, rho_x=1e-6, eps_abs=1e-4,
                    eps_rel=1e-4, eps_infeas=1e-7, alpha=1.5, cg_tol_override=cg_tol_override)
    for k, v in over.items():
        if k in ("verbose", "acceleration_lookback"):
            continue
        setattr(st, k, v)
    z, l, bsize, bl, bu, q, s = _cone_args(prob.cone)
    x, y, sv = _np.zeros(prob.n), _np.zeros(prob.m), _np.zeros(prob.m)
    info = OrInfo()
    cs = _np.ascontiguousarray(prob.cone.get("cs", []), dtype=_np.int32)
    pw = _np.ascontiguousarray(prob.cone.get("p", []), dtype=_np.float64)
    lib.or_solve_ext(prob.m, prob.n, _i(prob.Ap), _i(prob.Ai), _d(prob.Ax), _d(prob.b), _d(prob.c), z, l, bsize,
                     _d(bl), _d(bu), len(q), _i(q), len(s), _i(s), len(cs), _i(cs), int(prob.cone.get("ep", 0)),
                     int(prob.cone.get("ed", 0)), len(pw), _d(pw), C.byref(st), _d(x), _d(y), _d(sv),
                     C.byref(info))
    return dict(x=x, y=y, s=sv, info={k: getattr(info, k) for k, _ in OrInfo._fields_})
