import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
from scs_amd import capi
from oracle import pyoracle
ref = pyoracle.load_ref()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(5)
bu, bl = rng.uniform(0.1, 2.0, nb), -rng.uniform(0.1, 2.0, nb)
bu[::7] = 1e20
bl[::11] = -1e20
cone = dict(bu=bu, bl=bl)
r_y = rng.uniform(0.5, 3.0, nb + 1)
Tr = ref._scs_types
kr = capi.make_cone(cone, Tr)
wr = ref._scs_init_cone(C.byref(kr), nb + 1)
for mode in ("1", "0"):
    os.environ["SCS_AMD_BOX_MULTI"] = mode
    T = capi.T64
    lib = C.CDLL(capi.lib_path("libscsamd_cones.so"))
    lib._scs_init_cone.restype = C.c_void_p
    lib._scs_init_cone.argtypes = [C.POINTER(T.ScsCone), C.c_int]
    lib._scs_finish_cone.argtypes = [C.c_void_p]
    lib._scs_proj_dual_cone.argtypes = [T.fp, C.c_void_p, C.c_void_p, T.fp]
    k = capi.make_cone(cone)
    c = lib._scs_init_cone(C.byref(k), nb + 1)
    for rep in range(4):
        x0 = np.random.default_rng(100 + rep).standard_normal(nb + 1) * 2.0
        x0[0] = abs(x0[0]) * (0.2 if rep != 1 else -1.0)
        for r in (None, r_y):
            x = x0.copy()
            lib._scs_proj_dual_cone(x.ctypes.data_as(T.fp), c, None, r.ctypes.data_as(T.fp) if r is not None else None)
            want = x0.copy()
            if mode == "1":
                ref._scs_proj_dual_cone(want.ctypes.data_as(Tr.fp), wr, None, r.ctypes.data_as(Tr.fp) if r is not None else None)
                d = np.abs(x - want)
                print(f"mode {mode} rep {rep} r={r is not None}: t ours {x[0]:.15g} ref {want[0]:.15g} x0[0] {x0[0]:.6g} maxerr {d.max():.3e} at {d.argmax()} nbad {(d > 1e-9).sum()}", flush=True)
            else:
                print(f"mode {mode} rep {rep} r={r is not None}: t ours {x[0]:.15g}", flush=True)
    lib._scs_finish_cone(c)
