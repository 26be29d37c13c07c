"""Generate the golden fixtures under tests/golden/ by running the REAL reference
(oracle/_ref/*.so, built from /root/reference by oracle/Makefile) in this container.
The GPU box has no /root/reference; it only sees the committed .npz files.

    python tests/golden/make_golden.py

Fixtures:
  linsys_cfg1.npz   (b, s, tol) -> [x; y] pairs captured at the reference's
                    scs_solve_lin_sys boundary (oracle/trace_linsys.c) during a real
                    solve of BASELINE config 1 (n=1000, m=3000), with A and diag_r.
  cones.npz         x -> _scs_proj_dual_cone(x) pairs for zero/pos/box/SOC/PSD mixes,
                    Euclidean and R_y-weighted.
  solves.json       ScsInfo of full reference solves (with and without Anderson
                    acceleration) on seeded generator inputs.
"""
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from scs_amd import capi, problems  # noqa: E402
from oracle import pyoracle  # noqa: E402


def golden_linsys():
    calls = [0, 2, 3, 4, 30, 120]
    with tempfile.TemporaryDirectory() as d:
        os.environ["SCS_TRACE_DUMP"] = d
        os.environ["SCS_TRACE_CALLS"] = ",".join(map(str, calls))
        os.environ["SCS_TRACE_FILE"] = os.path.join(d, "trace.txt")
        ref = pyoracle.load_ref("libscsindir_ref_trace.so")
        pr = problems.random_socp(1000, 3000, 32, seed=1234)
        prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
        capi.solve(ref, prob, verbose=0, acceleration_lookback=0, max_iters=150, adaptive_scale=0)
        out = dict(n=1000, m=3000, Ap=prob.Ap, Ai=prob.Ai)
        out["Ax_normalized"] = np.fromfile(os.path.join(d, "A_x.bin"))
        out["diag_r"] = np.fromfile(os.path.join(d, "diag_r_init.bin"))
        tols = {}
        for line in open(os.path.join(d, "trace.txt")):
            f = line.split()
            if f[0].isdigit():
                tols[int(f[0])] = float(f[1].split("=")[1])
        for c in calls:
            out[f"b{c}"] = np.fromfile(os.path.join(d, f"call{c}_b.bin"))
            sp = os.path.join(d, f"call{c}_s.bin")
            out[f"s{c}"] = np.fromfile(sp) if os.path.exists(sp) else np.zeros(0)
            out[f"xy{c}"] = np.fromfile(os.path.join(d, f"call{c}_xy.bin"))
            out[f"tol{c}"] = tols[c]
        out["calls"] = np.array(calls)
        np.savez_compressed(os.path.join(HERE, "linsys_cfg1.npz"), **out)
        for k in ("SCS_TRACE_DUMP", "SCS_TRACE_CALLS", "SCS_TRACE_FILE"):
            os.environ.pop(k)


def ref_proj_dual(ref, cone, x, r_y=None):
    T = ref._scs_types
    k = capi.make_cone(cone, T)
    w = ref._scs_init_cone(C.byref(k), len(x))
    assert w
    out = np.array(x, dtype=np.float64)
    r = None if r_y is None else np.ascontiguousarray(r_y, dtype=np.float64)
    rc = ref._scs_proj_dual_cone(out.ctypes.data_as(T.fp), w, None, r.ctypes.data_as(T.fp) if r is not None else None)
    assert rc == 0, rc
    ref._scs_finish_cone(w)
    return out


def golden_cones():
    ref = pyoracle.load_ref()
    rng = np.random.default_rng(99)
    cases = {
        "zl": dict(z=7, l=13),
        "soc_small": dict(q=[1, 2, 3, 5, 4, 1, 9, 16]),
        "soc_mixed": dict(z=3, l=4, q=[17, 40, 2500, 3, 6000]),
        "box": dict(bl=(-rng.uniform(0.1, 2, 30)).tolist(), bu=rng.uniform(0.1, 2, 30).tolist()),
        "box_inf": dict(bl=[-1e20, -1.0, 0.0, -2.0], bu=[1.0, 1e20, 0.5, 3.0]),
        "psd_small": dict(s=[1, 2, 3, 4, 7]),
        "psd_50": dict(s=[50, 50, 33]),
        "mixed": dict(z=2, l=3, bl=[-1.0, -0.5], bu=[0.5, 2.0], q=[3, 20], s=[5, 12]),
        "exp": dict(ep=40, ed=35),
        "pow": dict(p=[0.5, 0.3, -0.25, 0.9, -0.7, 0.1, 0.5, -0.5] * 6),
        "all": dict(z=1, l=2, q=[4], s=[3], ep=5, ed=4, p=[0.4, -0.6, 0.8]),
        "cpsd": dict(cs=[1, 2, 3, 6, 11]),
        "all_c": dict(z=1, l=1, q=[3], s=[4], cs=[3, 2], ep=2, ed=1, p=[0.3]),
    }
    out = {}
    meta = {}
    for name, cone in cases.items():
        m = capi.cone_rows(cone)
        for variant in ("eucl", "ry"):
            x = rng.uniform(-2, 2, m)
            if name in ("exp", "pow"):
                x[: m // 2] *= 10.0 ** rng.uniform(-3, 2, m // 2)   # spread of magnitudes -> all branches
            if "psd_50" == name:
                x *= rng.uniform(0.1, 10)
            r_y = None
            if variant == "ry":
                # R_y as the solver builds it: constant within each cone of size > 1
                # (src/cones.c:349-363) -- here one value on the zero cone, another elsewhere,
                # and per-row values on the box cone, which is what the box metric sees
                r_y = np.full(m, 1.0 / 0.37)
                r_y[: cone.get("z", 0)] = 1.0 / (1000 * 0.37)
                zl = cone.get("z", 0) + cone.get("l", 0)
                nb = len(cone.get("bu", []))
                if nb:
                    r_y[zl: zl + nb + 1] = rng.uniform(0.5, 3.0, nb + 1)
            y = ref_proj_dual(ref, cone, x, r_y)
            out[f"{name}_{variant}_x"] = x
            out[f"{name}_{variant}_y"] = y
            if r_y is not None:
                out[f"{name}_{variant}_r"] = r_y
        meta[name] = cone
    np.savez_compressed(os.path.join(HERE, "cones.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "cones_meta.json"), "w"), indent=1)


def golden_solves():
    ref = pyoracle.load_ref()
    res = []
    specs = [
        dict(kind="socp", n=200, m=600, col_nnz=8, seed=1, over=dict(acceleration_lookback=0)),
        dict(kind="socp", n=1000, m=3000, col_nnz=32, seed=1234, over=dict(acceleration_lookback=0)),
        dict(kind="socp", n=1000, m=3000, col_nnz=32, seed=1234, over=dict()),  # AA on (default)
        dict(kind="socp", n=1000, m=3000, col_nnz=32, seed=1234, over=dict(acceleration_lookback=0, normalize=0)),
        dict(kind="socp", n=500, m=1500, col_nnz=6, seed=3, q_fixed=5, over=dict(acceleration_lookback=0)),
        dict(kind="sdp", n=60, n_blocks=6, block=8, bsize=21, col_nnz=6, seed=5, over=dict(acceleration_lookback=0)),
        dict(kind="sdp", n=60, n_blocks=6, block=8, bsize=21, col_nnz=6, seed=5, over=dict()),
    ]
    for sp in specs:
        if sp["kind"] == "socp":
            pr = problems.random_socp(sp["n"], sp["m"], sp["col_nnz"], seed=sp["seed"], q_fixed=sp.get("q_fixed"))
        else:
            pr = problems.random_sdp(sp["n"], sp["n_blocks"], sp["block"], sp["bsize"], sp["col_nnz"], seed=sp["seed"])
        prob = capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])
        r = capi.solve(ref, prob, verbose=0, **sp["over"])
        info = {k: (None if isinstance(v, float) and v != v else v) for k, v in r["info"].items()}
        res.append(dict(spec=sp, info=info, x_head=r["x"][:8].tolist(), x_absmax=float(np.abs(r["x"]).max()),
                        data_checksum=float(np.abs(prob.Ax).sum() + np.abs(prob.b).sum())))
    json.dump(res, open(os.path.join(HERE, "solves.json"), "w"), indent=1)


if __name__ == "__main__":
    golden_linsys()
    golden_cones()
    golden_solves()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
