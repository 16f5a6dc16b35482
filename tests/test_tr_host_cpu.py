"""The host half of row a14 without a GPU: mocap_ba_trust_region_step (csrc/tr_host.cpp + TrSubproblem, callable with
a NULL context) against scipy.optimize._lsq.common.solve_lsq_trust_region (scipy _lsq/trf.py:495 call site) on a
Jacobian of the ORACLE's restatement of the reference's residual_function (helpers.py:264-276), i.e. with the dead
focal-length columns the reference's parameter vector carries (helpers.py:247-262)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "low-cost-mocap_amd"))

from mocap_core import capi, helpers, synth  # noqa: E402
from oracle import mocap_oracle as mo  # noqa: E402


def _step(lib, JtJ, Jtr, m, Delta, alpha=0.0, method=0):
    n = JtJ.shape[0]
    JtJ = np.ascontiguousarray(JtJ, dtype=np.float64)
    Jtr = np.ascontiguousarray(Jtr, dtype=np.float64)
    a = ctypes.c_double(alpha)
    p = np.zeros(n)
    info = np.zeros(2, dtype=np.int32)
    rc = lib.mocap_ba_trust_region_step(None, n, int(m), JtJ.ctypes.data_as(ctypes.c_void_p), Jtr.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_double(Delta), ctypes.addressof(a), int(method),
                                        p.ctypes.data_as(ctypes.c_void_p), info.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return p, a.value, info


@pytest.mark.parametrize("C,N,seed", [(3, 40, 1), (4, 60, 2), (8, 90, 3)])
def test_trust_region_step_matches_scipy_on_the_oracle_jacobian(C, N, seed):
    from scipy.linalg import svd
    from scipy.optimize._lsq.common import solve_lsq_trust_region
    from scipy.optimize._numdiff import approx_derivative
    lib = capi.load_library()
    lib.mocap_ba_trust_region_step.restype = ctypes.c_int
    lib.mocap_ba_trust_region_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rig = synth.ring_rig(C)
    rng = np.random.default_rng(seed)
    obs, _ = synth.make_ba_observations(rig, N, seed=seed)
    init = synth.perturb_rig(rig, rng)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(C)])
    Ks = [k for k in rig["K"]]
    keep = ~np.isnan(np.asarray(mo.ba_residuals(x0, obs, Ks), dtype=np.float64))   # points the reference drops
    fun = lambda x: np.asarray(mo.ba_residuals(x, obs, Ks), dtype=np.float64)[keep]  # noqa: E731
    f = fun(x0)
    # (a step large enough to see through cv.projectPoints' float32 output: with scipy's default step the residual is
    # piecewise constant -- any matrix with the right zero columns would do for comparing the two subproblem solvers)
    J = approx_derivative(fun, x0, method="2-point", rel_step=1e-4, f0=f)
    m, n = J.shape
    dead = [0] + [1 + 7 * i for i in range(C - 1)]
    live = np.setdiff1d(np.arange(n), dead)
    assert not J[:, dead].any() and J[:, live].any(axis=0).all()      # the reference's dead focal parameters
    U, s, Vt = svd(J, full_matrices=False)
    uf = U.T @ f
    JtJ, Jtr = J.T @ J, J.T @ f
    # radii from just inside the Gauss-Newton step of the live block down to tiny.  (Beyond the Gauss-Newton step
    # scipy's rank-deficient branch divides the rounding dust LAPACK leaves in the null-space entries of `suf` by an
    # alpha of ~1e-17 and rescales the lot to Delta: that regime is compared on device-produced Jacobians, whose
    # exactly-zero columns come out of gesdd as exact zeros, in tests/test_gpu_ba.py.)
    Ul, sl, Vtl = svd(J[:, live], full_matrices=False)
    gn = np.linalg.norm(Vtl.T @ ((Ul.T @ f) / sl))
    for Delta in (0.9 * gn, 0.3 * gn, 0.01 * gn, 1e-4 * gn):
        for alpha0 in (0.0, 0.37):
            p_ref, a_ref, _ = solve_lsq_trust_region(n, m, uf, s, Vt.T, Delta, initial_alpha=alpha0)
            for method in (0, 1, 2):
                p, a, info = _step(lib, JtJ, Jtr, m, Delta, alpha0, method)
                assert int(info[1]) == len(live)
                assert not p[dead].any()
                np.testing.assert_allclose(a, a_ref, rtol=1e-7)
                np.testing.assert_allclose(p[live], p_ref[live], rtol=1e-6, atol=1e-7 * np.abs(p_ref).max())
                np.testing.assert_allclose(np.linalg.norm(p), Delta, rtol=1e-10)
