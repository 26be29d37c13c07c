"""The options entry (scs_amd/csrc/options.h): one table, one getenv, one programmatic entry.  The reference
has no environment variables (SURVEY section 5 "Config / flags"; include/scs.h:61-101 is its whole steering surface)."""
import glob
import os
import re

from scs_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scs_amd", "csrc")


def _sources():
    for pat in ("*.h", "*.hip", "*.cpp"):
        for f in sorted(glob.glob(os.path.join(CSRC, pat))):
            yield f, open(f).read()


def test_the_environment_is_read_in_one_place_only():
    offenders = [os.path.basename(f) for f, s in _sources() if os.path.basename(f) != "options.h" and re.search(r"\bgetenv\s*\(", s)]
    assert offenders == [], offenders
    for py in ("capi.py", "solver.py", "batch.py", "shard.py", "problems.py", "verify.py"):
        s = open(os.path.join(ROOT, "scs_amd", py)).read()
        assert "SCS_AMD_" not in s, py  # the Python mirror steers through set_option, not through variables of its own


def test_every_option_used_in_the_sources_has_a_row_and_every_row_is_documented():
    lib = capi.load("libscsamd.so")
    rows = {r["key"]: r for r in capi.list_options(lib)}
    used = set()
    for f, s in _sources():
        used |= set(re.findall(r"opt_(?:get|is_set)\(\s*\"([a-z0-9_]+)\"\s*\)", s))
    assert used, "no option reads found: the pattern of this test is stale"
    assert used - set(rows) == set(), sorted(used - set(rows))
    assert set(rows) - used == set(), ("rows nothing reads", sorted(set(rows) - used))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- options-table-begin -->(.*?)<!-- options-table-end -->", doc, re.S)
    assert m, "INTEGRATION.md section 5 has lost its option table"
    documented = {}
    for line in m.group(1).splitlines():
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) >= 5 and c[0].startswith("`"):
            documented[c[0].strip("`")] = c
    assert set(documented) == set(rows), (sorted(set(rows) - set(documented)), sorted(set(documented) - set(rows)))
    for k, r in rows.items():
        assert documented[k][1] == r["cls"], k
        assert documented[k][2].startswith({0: "no", 1: "rounding", 2: "trajectory"}[r["numerics"]]), k
        assert r["cls"] in ("supported", "ab", "test", "diag")
    # every library carries the same table
    import ctypes as C
    for name in ("libscsamd_linsys.so", "libscsamd_cones.so", "libscsamd_f32.so", "libscsamd_dlong.so"):
        other = C.CDLL(capi.lib_path(name))  # the cone library exports the nine _scs_* symbols only: bound by hand
        other.scs_amd_list_options.restype = C.c_longlong if "dlong" in name else C.c_int
        other.scs_amd_list_options.argtypes = [C.c_char_p, other.scs_amd_list_options.restype]
        assert {r["key"] for r in capi.list_options(other)} == set(rows), name


def test_set_get_and_the_gated_environment_fallback(monkeypatch):
    lib = capi.load("libscsamd_linsys.so")
    for k in ("SCS_AMD_REORDER", "SCS_AMD_CG3", "SCS_AMD_ALLOW_ENV_HOOKS"):
        monkeypatch.delenv(k, raising=False)
    assert lib.scs_amd_set_option(b"no_such_switch", b"1") == -1
    assert lib.scs_amd_get_option(b"no_such_switch") is None
    assert lib.scs_amd_get_option(b"reorder") is None
    # programmatic value wins over the environment; NULL restores the fallback
    monkeypatch.setenv("SCS_AMD_REORDER", "0")
    assert lib.scs_amd_get_option(b"reorder") == b"0"          # class supported: environment honoured
    assert lib.scs_amd_set_option(b"reorder", b"2") == 0
    assert lib.scs_amd_get_option(b"reorder") == b"2"
    assert lib.scs_amd_set_option(b"reorder", None) == 0
    assert lib.scs_amd_get_option(b"reorder") == b"0"
    # measurement variants / test hooks: a stray variable does nothing ...
    monkeypatch.setenv("SCS_AMD_CG3", "1")
    assert lib.scs_amd_get_option(b"cg3") is None
    monkeypatch.setenv("SCS_AMD_ALLOW_ENV_HOOKS", "0")
    assert lib.scs_amd_get_option(b"cg3") is None
    # ... unless the caller says it is running a test / an A/B measurement, or sets it through the entry
    monkeypatch.setenv("SCS_AMD_ALLOW_ENV_HOOKS", "1")
    assert lib.scs_amd_get_option(b"cg3") == b"1"
    monkeypatch.delenv("SCS_AMD_ALLOW_ENV_HOOKS")
    assert lib.scs_amd_set_option(b"cg3", b"1") == 0
    assert lib.scs_amd_get_option(b"cg3") == b"1"
    assert lib.scs_amd_set_option(b"cg3", None) == 0
    assert lib.scs_amd_get_option(b"cg3") is None
