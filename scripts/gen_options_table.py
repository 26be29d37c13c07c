"""Prints the option table of scs_amd/csrc/options.h as the markdown block INTEGRATION.md section 5 carries (between the
options-table markers); tests/test_options.py compares the two.  Usage: python scripts/gen_options_table.py [--write]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scs_amd import capi  # noqa: E402


def table():
    num = {0: "no (bit-identical)", 1: "rounding (other summation / rotation order)", 2: "trajectory"}
    out = ["| key (`scs_amd_set_option`; env `SCS_AMD_<KEY>`) | class | changes numerics | values | meaning |", "|---|---|---|---|---|"]
    for r in capi.list_options():
        out.append(f"| `{r['key']}` | {r['cls']} | {num[r['numerics']]} | `{r['values'].replace('|', ' / ')}` | {r['doc'].replace('|', '/')} |")
    return "\n".join(out)


if __name__ == "__main__":
    t = table()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "INTEGRATION.md")
        s = open(p).read()
        s2 = re.sub(r"(<!-- options-table-begin -->\n).*?(<!-- options-table-end -->)", lambda m: m.group(1) + t + "\n" + m.group(2), s, flags=re.S)
        open(p, "w").write(s2)
    else:
        print(t)
