"""The C-ABI libraries load and export every symbol include/scs_amd.h declares;
struct layouts in the ctypes binding equal the C compiler's view of the header.
No compute calls (CPU only)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from scs_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "scs_amd.h")


def _declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scs_[a-z_0-9]*|scs)\s*\(", src)) - {"scs_int", "scs_float"})


def _exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.lib_path(lib)], text=True)
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def test_full_library_exports_every_declared_symbol():
    fns = _declared_functions()
    assert "scs_init" in fns and "scs_solve_lin_sys" in fns and "scs_amd_cone_proj_dual" in fns and "scs" in fns
    for lib in ("libscsamd.so", "libscsamd_f32.so", "libscsamd_dlong.so"):
        exp = _exported(lib)
        missing = [f for f in fns if f not in exp]
        assert not missing, (lib, missing)
        C.CDLL(capi.lib_path(lib))  # loads (HIP runtime resolves) even without a GPU


def test_linsys_plugin_library_exports_only_the_plugin():
    exp = _exported("libscsamd_linsys.so")
    five = {"scs_init_lin_sys_work", "scs_solve_lin_sys", "scs_update_lin_sys_diag_r", "scs_free_lin_sys_work",
            "scs_get_lin_sys_method"}
    assert five <= exp
    assert all(s in five or s.startswith("scs_amd_") for s in exp), exp


def test_no_gpu_is_reported_not_faked():
    lib = capi.load("libscsamd.so")
    n = lib.scs_amd_device_count()
    assert isinstance(n, int)
    assert lib.scs_get_lin_sys_method().decode().startswith("sparse-indirect")
    assert b"amd" in lib.scs_version()


@pytest.mark.parametrize("flag,T", [("", capi.T64), ("-DSFLOAT=1", capi.T32), ("-DDLONG=1", capi.T64L)])
def test_ctypes_structs_match_the_header(flag, T):
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "scs_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ScsMatrix), sizeof(ScsSettings), sizeof(ScsData), sizeof(ScsCone),
         sizeof(ScsSolution), sizeof(AaStats), sizeof(ScsInfo), sizeof(ScsAmdStats));
  printf("%zu %zu %zu %zu\n", offsetof(ScsInfo, status_val), offsetof(ScsInfo, aa_stats), offsetof(ScsInfo, lin_sys_time),
         offsetof(ScsSettings, write_data_filename));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include")] + ([flag] if flag else []) + [src, "-o", exe])
        l1, l2 = subprocess.check_output([exe], text=True).strip().splitlines()
    sizes = [int(v) for v in l1.split()]
    want = [C.sizeof(t) for t in (T.ScsMatrix, T.ScsSettings, T.ScsData, T.ScsCone, T.ScsSolution, T.AaStats,
                                  T.ScsInfo, T.ScsAmdStats)]
    assert sizes == want
    offs = [int(v) for v in l2.split()]
    assert offs == [T.ScsInfo.status_val.offset, T.ScsInfo.aa_stats.offset, T.ScsInfo.lin_sys_time.offset,
                    T.ScsSettings.write_data_filename.offset]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "LIB_DIR", str(tmp_path))
    capi._cache.pop("libscsamd.so", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load("libscsamd.so")
