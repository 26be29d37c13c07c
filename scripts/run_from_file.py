#!/usr/bin/env python3
"""Replay a problem dump in the reference's binary layout on the GPU -- the counterpart of
reference test/run_from_file.c (`run_from_file <file> [max_iters] [eps]`)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scs_amd import capi

lib = capi.load("libscsamd.so")
T = lib._scs_types
PD, PK, PS = C.POINTER(T.ScsData), C.POINTER(T.ScsCone), C.POINTER(T.ScsSettings)
lib.scs_amd_read_data.restype = C.c_int
lib.scs_amd_read_data.argtypes = [C.c_char_p, C.POINTER(PD), C.POINTER(PK), C.POINTER(PS)]
lib.scs_amd_free_data.argtypes = [PD, PK, PS]
d, k, s = PD(), PK(), PS()
if lib.scs_amd_read_data(sys.argv[1].encode(), C.byref(d), C.byref(k), C.byref(s)) != 0:
    raise SystemExit(1)
if len(sys.argv) > 2:
    s.contents.max_iters = int(sys.argv[2])
if len(sys.argv) > 3:
    s.contents.eps_abs = s.contents.eps_rel = float(sys.argv[3])
sol, info = T.ScsSolution(), T.ScsInfo()
lib.scs(d, k, s, C.byref(sol), C.byref(info))
print(json.dumps(capi.info_dict(info)))
lib.scs_amd_free_data(d, k, s)
