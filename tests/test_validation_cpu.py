"""Input validation of the B2 entry points happens on the host before any device work, so
the error behaviour of reference src/scs.c:376-449 / linsys/scs_matrix.c:65-157 /
src/cones.c:583-700 is checked here without a GPU: every bad input makes scs_init return
NULL (never a crash, never a silent fix-up)."""
import ctypes as C

import numpy as np
import pytest

from scs_amd import capi, problems


@pytest.fixture(scope="module")
def lib():
    return capi.load("libscsamd.so")


def _prob(n=6, m=10):
    pr = problems.random_cone_prob(n, m, 2, dict(z=2, l=3, q=[5]), seed=1)
    return capi.Problem(pr["A"], pr["b"], pr["c"], pr["cone"])


def _init(lib, prob, **over):
    st = capi.default_settings(lib, verbose=0, **over)
    return lib.scs_init(C.byref(prob.data), C.byref(prob.k), C.byref(st))


def test_defaults_match_reference_glbopts(lib):
    st = capi.default_settings(lib)   # include/glbopts.h:35-50
    assert (st.max_iters, st.eps_abs, st.eps_rel, st.eps_infeas) == (100000, 1e-4, 1e-4, 1e-7)
    assert (st.alpha, st.rho_x, st.scale, st.normalize, st.adaptive_scale) == (1.5, 1e-6, 0.1, 1, 1)
    assert (st.acceleration_lookback, st.acceleration_interval, st.acceleration_type_1) == (10, 10, 1)
    assert (st.acceleration_regularization, st.acceleration_relaxation) == (1e-8, 1.0)


@pytest.mark.parametrize("over", [dict(max_iters=0), dict(eps_abs=-1.0), dict(eps_rel=float("nan")), dict(alpha=2.0),
                                  dict(alpha=0.0), dict(rho_x=0.0), dict(scale=-1.0), dict(time_limit_secs=-1.0),
                                  dict(acceleration_interval=0), dict(acceleration_lookback=-1),
                                  dict(acceleration_relaxation=2.5), dict(acceleration_regularization=-1.0)])
def test_bad_settings_are_rejected(lib, over):
    assert not _init(lib, _prob(), **over)


def test_null_inputs(lib):
    prob = _prob()
    st = capi.default_settings(lib, verbose=0)
    assert not lib.scs_init(None, C.byref(prob.k), C.byref(st))
    assert not lib.scs_init(C.byref(prob.data), None, C.byref(st))
    assert not lib.scs_init(C.byref(prob.data), C.byref(prob.k), None)


def test_cone_dimension_mismatch_and_unsupported_cones(lib):
    prob = _prob()
    prob.k.l += 1                       # rows no longer add up to m
    assert not _init(lib, prob)
    prob = _prob()
    prob.k.cssize = 1                   # complex PSD cone count without its size array
    assert not _init(lib, prob)
    prob = _prob()
    prob.k.ep = 1                       # 3 more rows than A has
    assert not _init(lib, prob)
    prob = _prob()
    q = np.array([-5], dtype=np.int32)
    prob.k.q = q.ctypes.data_as(capi.T64.ip)
    assert not _init(lib, prob)


def test_bad_matrix_is_rejected(lib):
    prob = _prob()
    prob.Ai[0] = prob.m + 3             # row index out of range
    assert not _init(lib, prob)
    prob = _prob()
    prob.Ax[1] = np.inf
    assert not _init(lib, prob)
    prob = _prob()
    prob.Ap[1], prob.Ap[2] = prob.Ap[2], prob.Ap[1] - 1 if prob.Ap[1] else 0   # non-monotone
    prob.Ap[2] = 0
    assert not _init(lib, prob)


def test_box_bounds_inverted(lib):
    cone = dict(bl=[1.0, -1.0], bu=[0.0, 1.0])
    pr = problems.random_cone_prob(4, 3, 2, dict(bl=[-1.0, -1.0], bu=[1.0, 1.0]), seed=2)
    prob = capi.Problem(pr["A"], pr["b"], pr["c"], cone)
    assert not _init(lib, prob)


def test_solve_with_null_arguments_fails_cleanly(lib):
    T = lib._scs_types
    info = T.ScsInfo()
    assert lib.scs_solve(None, None, C.byref(info), 0) == -4   # SCS_FAILED


def test_python_solver_object_reports_missing_gpu_or_bad_data():
    """scs_amd.solver.SCS (the scs-python-shaped object): malformed data is rejected before any
    device work; without a GPU the constructor fails loudly (no CPU fallback)."""
    import numpy as np
    import scipy.sparse as sp
    from scs_amd.solver import SCS
    A = sp.csc_matrix(np.eye(3))
    with pytest.raises(ValueError):
        SCS(dict(A=A, b=np.ones(3)), dict(l=3))
    lib_ = capi.load("libscsamd.so")
    if lib_.scs_amd_device_count() <= 0:
        with pytest.raises(ValueError, match="ScsWork allocation error"):
            SCS(dict(A=A, b=np.ones(3), c=np.ones(3)), dict(l=3))
