"""`SCS` -- the object scs-python users hold, over this library's C ABI.

Mirrors the reference project's Python interface (docs/src/api/python.rst in the
reference tree: `scs.SCS(data, cone, **settings)`, `.solve(warm_start=True, x=None, y=None,
s=None)`, `.update(b=None, c=None)`, result dict with 'x', 'y', 's', 'info') so that code
written against `import scs` switches with `from scs_amd.solver import SCS`.  One ScsWork
lives for the lifetime of the object (scs_init once, scs_solve / scs_update many times,
scs_finish on close / garbage collection), exactly the workspace reuse the C API offers
(include/scs.h:271-324).  Host code only: every flop is in libscsamd.so.

data : dict with 'A' (scipy sparse, m x n), 'b' (m), 'c' (n), optional 'P' (n x n, symmetric;
       the upper triangle is used, as the C API requires)
cone : dict with any of 'z' (or legacy 'f'), 'l', 'bu', 'bl', 'q', 's', 'cs', 'ep', 'ed', 'p'
settings : the fields of ScsSettings (eps_abs, max_iters, acceleration_lookback, ...);
       `use_indirect` / `gpu` / `mkl` are accepted and ignored (there is one backend);
       `dtype="f32"` selects the SFLOAT library.
"""
import ctypes as C

import numpy as np

from . import capi

_IGNORED = {"use_indirect", "gpu", "mkl", "linear_solver"}


class SCS:
    def __init__(self, data, cone, dtype="f64", **settings):
        self._lib = capi.load("libscsamd_f32.so" if dtype in ("f32", np.float32) else "libscsamd.so")
        self._T = self._lib._scs_types
        self._w = None
        if any(k not in data for k in ("A", "b", "c")):
            raise ValueError("data must contain 'A', 'b' and 'c'")
        cone = dict(cone)
        if "f" in cone:  # scs < 3 name of the zero cone
            cone["z"] = cone.pop("f") + cone.get("z", 0)
        for k in ("q", "s", "cs", "p", "bu", "bl"):  # scs-python accepts scalars for single cones
            if k in cone and np.isscalar(cone[k]):
                cone[k] = [cone[k]]
        self._prob = capi.Problem(data["A"], data["b"], data["c"], cone, P=data.get("P"), T=self._T)
        kw = {k: v for k, v in settings.items() if k not in _IGNORED}
        for k in ("write_data_filename", "log_csv_filename"):
            if isinstance(kw.get(k), str):
                kw[k] = kw[k].encode()
        kw.setdefault("verbose", 0)
        self._settings = capi.default_settings(self._lib, **kw)
        self._w = self._lib.scs_init(C.byref(self._prob.data), C.byref(self._prob.k), C.byref(self._settings))
        if not self._w:
            raise ValueError("ScsWork allocation error!")  # scs-python's message for a failed scs_init
        f = self._T.np_float
        self._x = np.zeros(self._prob.n, dtype=f)
        self._y = np.zeros(self._prob.m, dtype=f)
        self._s = np.zeros(self._prob.m, dtype=f)
        self._solved_once = False

    def solve(self, warm_start=True, x=None, y=None, s=None):
        """scs_solve.  With warm_start the previous solution (or the x / y / s given) seeds the
        iteration, as in scs-python; the first solve of an object without x / y / s is cold."""
        if not self._w:
            raise RuntimeError("solver was closed")
        T = self._T
        given = [v is not None for v in (x, y, s)]
        for dst, src in ((self._x, x), (self._y, y), (self._s, s)):
            if src is not None:
                src = np.asarray(src, dtype=dst.dtype)
                if src.shape != dst.shape:
                    raise ValueError("warm-start vector has the wrong length")
                dst[:] = src
        warm = bool(warm_start) and (self._solved_once or any(given))
        sol = T.ScsSolution(self._x.ctypes.data_as(T.fp), self._y.ctypes.data_as(T.fp), self._s.ctypes.data_as(T.fp))
        info = T.ScsInfo()
        self._lib.scs_solve(self._w, C.byref(sol), C.byref(info), 1 if warm else 0)
        self._solved_once = True
        return {"x": self._x.copy(), "y": self._y.copy(), "s": self._s.copy(), "info": capi.info_dict(info)}

    def update(self, b=None, c=None):
        """scs_update: new right-hand side and / or cost on the same factorised workspace."""
        if not self._w:
            raise RuntimeError("solver was closed")
        T = self._T
        f = T.np_float
        bp = cp = None
        if b is not None:
            self._b_new = np.ascontiguousarray(b, dtype=f)
            if self._b_new.shape != (self._prob.m,):
                raise ValueError("b has the wrong length")
            bp = self._b_new.ctypes.data_as(T.fp)
        if c is not None:
            self._c_new = np.ascontiguousarray(c, dtype=f)
            if self._c_new.shape != (self._prob.n,):
                raise ValueError("c has the wrong length")
            cp = self._c_new.ctypes.data_as(T.fp)
        if self._lib.scs_update(self._w, bp, cp) != 0:
            raise RuntimeError("scs_update failed")

    def close(self):
        if getattr(self, "_w", None):
            self._lib.scs_finish(self._w)
            self._w = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve(data, cone, **settings):
    """One-shot form (scs-python 2.x `scs.solve`)."""
    with SCS(data, cone, **settings) as solver:
        return solver.solve()
