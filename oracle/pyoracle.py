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
