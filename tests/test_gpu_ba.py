"""GPU: bundle-adjustment pieces of the HIP core against the reference goldens / oracle / SciPy.

Parity ladder (a13/a14 of SURVEY.md section 8):
  (1) the residual vector (float64 before the cast, and the float32 cast of helpers.py:273),
  (2) J^T J / J^T f from the MFMA contraction vs NumPy on the same J, and one float32-flow linearisation
      against SciPy's own approx_derivative / loss / scaling functions,
  (3) the trust-region step vs scipy's solve_lsq_trust_region on the same J, f, Delta,
  (4) the reference's own bundle_adjustment results (tests/golden/ba_*solved: poses + OptimizeResult
      statistics produced by the reference run through the stub harness) vs mode "scipy" (reference
      optimizer, GPU residuals) and mode "resident" (mocap_ba_solve),
  (5) final poses in tight mode (float64 residuals, tolerances 1e-12) vs scipy.least_squares driven
      by the oracle's residuals -- both must reach the same minimum to 1e-5 relative.
"""
import os

import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_names("ba_"))
def test_residuals_vs_reference_golden(core, name):
    g = load_golden(name)
    core.set_cameras(g["K"], g["R_init"], g["t_init"])
    r = core.ba_residuals(g["xs"], g["obs"])
    for p in range(len(g["xs"])):
        valid = np.isfinite(r[p])
        assert valid.sum() == g["res64"].shape[1]
        np.testing.assert_allclose(r[p][valid], g["res64"][p], rtol=1e-5, atol=0)      # contract
        np.testing.assert_allclose(r[p][valid], g["res64"][p], rtol=1e-3 * 1e-3)      # achieved (<=1e-6)
        # the reference's float32 cast: equal except where a float32 rounding boundary is crossed
        same = r[p][valid].astype(np.float32) == g["res32"][p]
        assert same.mean() > 0.95


def test_residuals_vs_c_oracle_1k_points(core):
    from mocap_core import synth
    from oracle import c_oracle
    rig = synth.ring_rig(8)
    rng = np.random.default_rng(61)
    obs, _ = synth.make_ba_observations(rig, 1000, seed=61)
    init = synth.perturb_rig(rig, rng)
    core.set_cameras(rig["K"], init["R"], init["t"])
    from mocap_core import helpers
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
    xs = np.stack([x0] + [x0 + rng.normal(0, 1e-3, x0.size) for _ in range(4)])
    r = core.ba_residuals(xs, obs)
    ref = c_oracle.COracle(rig["K"], init["R"], init["t"]).ba_residuals(xs, obs)
    assert np.array_equal(np.isnan(r), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(r[ok], ref[ok], rtol=1e-3, atol=1e-12)   # float32-boundary flips, see test_gpu_parity
    assert np.median(np.abs(r[ok] - ref[ok]) / ref[ok]) < 1e-9


def test_gram_mfma_vs_numpy(core):
    """J^T J and J^T f from v_mfma_f64_16x16x4_f64 against NumPy on the very same (scaled) J."""
    from mocap_core import helpers, synth
    rig = synth.ring_rig(8)
    rng = np.random.default_rng(62)
    obs, _ = synth.make_ba_observations(rig, 1000, seed=62)
    init = synth.perturb_rig(rig, rng)
    core.set_cameras(rig["K"], init["R"], init["t"])
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
    for f32 in (False, True):
        ne = core.ba_normal_eq(x0, obs, f32_residuals=f32, use_cauchy=True, want_J=True)
        J = ne["J"]
        assert J.shape == (ne["m"], 50)
        # J is asymmetric in rows/cols, so a row<->col or row-mapping slip in the MFMA output shows up
        np.testing.assert_allclose(ne["JtJ"], J.T @ J, rtol=1e-12, atol=1e-12 * np.abs(J.T @ J).max())
        # dead focal parameters: exactly zero columns (helpers.py:267-270)
        dead = [0] + [1 + 7 * i for i in range(7)]
        assert not J[:, dead].any() and not ne["JtJ"][dead].any()
        # gradient and cost against the oracle-free definition (float64 flow; the float32 flow is the next test)
        r0 = core.ba_residuals(x0, obs)[0]
        f = r0[~np.isnan(r0)]
        if f32:
            continue
        z = f * f
        rho1 = 1 / (1 + z)
        jscale = np.sqrt(np.maximum(rho1 - 2 * z / (1 + z) ** 2, np.finfo(float).eps))
        np.testing.assert_allclose(ne["cost"], 0.5 * np.log1p(z).sum(), rtol=1e-12)
        np.testing.assert_allclose(ne["Jtr"], J.T @ (f * rho1 / jscale), rtol=1e-10, atol=1e-10 * np.abs(ne["Jtr"]).max())


@pytest.mark.parametrize("C,N,seed", [(4, 200, 75), (8, 1000, 76)])
def test_float32_linearisation_is_scipys_own_arithmetic(core, C, N, seed):
    """Reference settings (helpers.py:273: float32 residuals): one linearisation of the resident loop against
    SciPy's OWN functions fed with float32 residuals -- approx_derivative (float32 differences, float64
    quotient), construct_loss_function('cauchy') (float32 z, rho), scale_for_robust_loss_function (float64
    J_scale, float32 scaled f).  NumPy's promotion rules decide every intermediate precision; the kernel
    restates them operation by operation, so J agrees to rounding of the last float64 operation, not to 1e-7."""
    from scipy.optimize._lsq.common import scale_for_robust_loss_function
    from scipy.optimize._lsq.least_squares import construct_loss_function
    from scipy.optimize._numdiff import approx_derivative
    from mocap_core import helpers, synth
    rig = synth.ring_rig(C)
    rng = np.random.default_rng(seed)
    obs, _ = synth.make_ba_observations(rig, N, seed=seed, dropout=0.1)
    init = synth.perturb_rig(rig, rng)
    core.set_cameras(rig["K"], init["R"], init["t"])
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(C)])

    def fun(x):                                                   # = the reference's residual_function
        r = core.ba_residuals(x, obs)[0]
        return r[~np.isnan(r)].astype(np.float32)

    f0 = fun(x0)
    J_ref = approx_derivative(fun, x0, method="2-point", f0=f0)   # least_squares.py:903-906
    loss = construct_loss_function(f0.size, "cauchy", 1.0)
    rho = loss(f0)                                                # trf.py: rho = loss_function(f)
    cost_ref = 0.5 * np.sum(rho[0])
    f_ref = f0.copy()
    J_ref, f_ref = scale_for_robust_loss_function(J_ref, f_ref, rho)
    assert f_ref.dtype == np.float32 and J_ref.dtype == np.float64
    ne = core.ba_normal_eq(x0, obs, f32_residuals=True, use_cauchy=True, want_J=True)
    assert ne["m"] == f0.size
    np.testing.assert_allclose(ne["J"], J_ref, rtol=1e-13, atol=1e-13 * np.abs(J_ref).max())
    # float32 log1p: the kernel rounds correctly, NumPy's SIMD loop may differ in the last float32 bit
    np.testing.assert_allclose(ne["cost"], cost_ref, rtol=2e-7)
    g_ref = J_ref.T.dot(f_ref)                                    # compute_grad
    np.testing.assert_allclose(ne["Jtr"], g_ref, rtol=1e-11, atol=1e-12 * np.abs(g_ref).max())
    np.testing.assert_allclose(ne["JtJ"], J_ref.T @ J_ref, rtol=1e-11, atol=1e-12 * np.abs(J_ref.T @ J_ref).max())


def test_jacobian_vs_scipy_numdiff(core):
    """The batched forward differences equal scipy's approx_derivative driven by the same residuals."""
    from scipy.optimize._numdiff import approx_derivative
    from mocap_core import helpers, synth
    rig = synth.ring_rig(4)
    rng = np.random.default_rng(63)
    obs, _ = synth.make_ba_observations(rig, 200, seed=63)
    init = synth.perturb_rig(rig, rng)
    core.set_cameras(rig["K"], init["R"], init["t"])
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(4)])

    def fun(x):
        r = core.ba_residuals(x, obs)[0]
        return r[~np.isnan(r)]

    ne = core.ba_normal_eq(x0, obs, f32_residuals=False, use_cauchy=False, want_J=True)
    J_ref = approx_derivative(fun, x0, method="2-point")
    np.testing.assert_allclose(ne["J"], J_ref, rtol=1e-12, atol=1e-12 * np.abs(J_ref).max())


def _linearised(core, C, N, seed):
    """(J, scaled f, x0) of one linearisation on the GPU -- the inputs scipy's trf loop hands to its
    subproblem solver (scipy _lsq/trf.py:433-446: J and f after scale_for_robust_loss_function)."""
    from mocap_core import helpers, synth
    rig = synth.ring_rig(C)
    rng = np.random.default_rng(seed)
    obs, _ = synth.make_ba_observations(rig, N, seed=seed)
    init = synth.perturb_rig(rig, rng)
    core.set_cameras(rig["K"], init["R"], init["t"])
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(C)])
    ne = core.ba_normal_eq(x0, obs, f32_residuals=False, use_cauchy=True, want_J=True)
    r0 = core.ba_residuals(x0, obs)[0]
    f = r0[~np.isnan(r0)]
    z = f * f
    f_scaled = f * (1 / (1 + z)) / np.sqrt(np.maximum(1 / (1 + z) - 2 * z / (1 + z) ** 2, np.finfo(float).eps))
    return ne, f_scaled, x0


@pytest.mark.parametrize("C,N,seed,well_conditioned", [(3, 60, 71, False), (4, 300, 64, True), (4, 300, 72, False),
                                                      (8, 1000, 73, True), (8, 1000, 74, True)])
def test_trust_region_step_vs_scipy(core, C, N, seed, well_conditioned):
    """The subproblem solution itself (step p and Levenberg-Marquardt parameter alpha) of mocap_ba_solve's
    trust-region solver against scipy.optimize._lsq.common.solve_lsq_trust_region on the same J, f, Delta
    (scipy _lsq/trf.py:495 call site; J = U S V^T by SVD as trf.py:448 does).  No LM loop in between, so
    there is nothing to reject and nothing to skip.  Radii from "Gauss-Newton step far outside" to "inside"."""
    from scipy.linalg import svd
    from scipy.optimize._lsq.common import solve_lsq_trust_region
    ne, f_scaled, x0 = _linearised(core, C, N, seed)
    J = ne["J"]
    m, n = J.shape
    U, s, Vt = svd(J, full_matrices=False)
    uf = U.T @ f_scaled
    dead = [0] + [1 + 7 * i for i in range(C - 1)]
    live = np.setdiff1d(np.arange(n), dead)
    # the LM loop hands the solver J^T J and J^T f from the MFMA contraction, so that is what is passed here
    JtJ, Jtr = ne["JtJ"], ne["Jtr"]
    np.testing.assert_allclose(Jtr, J.T @ f_scaled, rtol=1e-9, atol=1e-9 * np.abs(Jtr).max())
    x_norm = np.linalg.norm(x0)
    for Delta in (x_norm, 0.1 * x_norm, 1e-2, 1e-4):
        for alpha0 in (0.0, 0.37):
            p_ref, a_ref, _ = solve_lsq_trust_region(n, m, uf, s, Vt.T, Delta, initial_alpha=alpha0)
            for method in (0, 1, 2):
                p, a, info = core.ba_trust_region_step(JtJ, Jtr, m, Delta, alpha=alpha0, method=method)
                assert info["live"] == n - len(dead)
                assert info["method"] == (1 if method == 1 else 2)      # dead columns -> rank-deficient branch
                # dead parameters get exactly 0 here; scipy's SVD leaves rounding dust there (up to ~1e-8 of
                # the step at large radii, measured) -- without effect, the parameters are dead
                assert not p[dead].any()
                np.testing.assert_allclose(a, a_ref, rtol=1e-8)         # measured: <= 4e-11
                np.testing.assert_allclose(p[live], p_ref[live], rtol=1e-7, atol=1e-8 * np.abs(p_ref).max())  # <= 2e-10
                np.testing.assert_allclose(np.linalg.norm(p), Delta, rtol=1e-12)
    # full-rank branch (scipy's `full_rank`; never taken by the reference, whose J always has dead columns): the
    # live block alone, radius large enough for the Gauss-Newton step and then small enough to bind.  The eigen
    # path works on J^T J, i.e. on cond(J)^2, so the unregularised step is compared where cond(J)^2 eps << 1:
    # seeds 64 / 73 / 74 have cond ~ 30; seeds 71 / 72 drew a rig whose scale gauge is nearly free (cond 7e7 and
    # 1e7: the Gauss-Newton step itself is 1e10 long there) and are covered by the regularised branch above.
    if not well_conditioned:
        assert s[len(live) - 1] / s[0] < 1e-6
        return
    assert s[0] / s[len(live) - 1] < 1e3
    Jl = J[:, live]
    Ul, sl, Vtl = svd(Jl, full_matrices=False)
    ufl = Ul.T @ f_scaled
    gn = np.linalg.norm(Vtl.T @ (ufl / sl))
    for Delta in (2.0 * gn, 0.5 * gn, 0.01 * gn):
        p_ref, a_ref, _ = solve_lsq_trust_region(len(live), m, ufl, sl, Vtl.T, Delta, initial_alpha=0.0)
        for method in (0, 1):
            p, a, info = core.ba_trust_region_step(Jl.T @ Jl, Jl.T @ f_scaled, m, Delta, method=method)
            assert info["method"] == 1 and info["live"] == len(live)
            np.testing.assert_allclose(a, a_ref, rtol=1e-8, atol=1e-18)
            np.testing.assert_allclose(p, p_ref, rtol=1e-8, atol=1e-10 * np.abs(p_ref).max())   # measured: 3e-14


@pytest.mark.parametrize("C,N,budget", [(4, 200, 400), (8, 1000, 120)])
def test_solve_tight_mode_matches_scipy(core, C, N, budget):
    """Tight mode (all-double arithmetic, float64 residuals, tolerances 1e-12, noise-free captures): the
    resident LM loop and scipy.optimize.least_squares fed by the ORACLE's residuals follow the same
    trust-region trajectory -- same evaluation count, same cost, poses equal to <= 1e-5 -- and, where the
    budget lets them converge (4 cameras), both land on the generating rig.
    OpenCV's float32 roundings must be off here: with them the residual is piecewise constant at the
    1e-5 px level and a 1.5e-8 finite-difference step measures only quantisation noise (both solvers
    then stall on the same ~0.3 cost plateau).  The residual is a *squared* error, so Gauss-Newton
    converges only linearly near the minimum: 8 cameras do not get there in a test-sized budget."""
    from scipy import optimize
    from mocap_core import helpers, synth
    from oracle import c_oracle
    rig = synth.ring_rig(C)
    rng = np.random.default_rng(65 + C)
    obs, _ = synth.make_ba_observations(rig, N, seed=65 + C, noise_px=0.0)
    init = synth.perturb_rig(rig, rng, rot_sigma=0.01, trans_sigma=0.02)
    core.set_cameras(rig["K"], init["R"], init["t"])
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(C)])
    core.set_options(f32_rounding=False)
    try:
        x_gpu, info = core.ba_solve(x0, obs, ftol=1e-12, xtol=1e-12, gtol=1e-12, max_iter=budget, f32_residuals=False)
    finally:
        core.set_options(f32_rounding=True)
    co = c_oracle.COracle(rig["K"], init["R"], init["t"], f32_rounding=False)

    def fun(x):
        r = co.ba_residuals(x, obs)[0]
        return r[~np.isnan(r)]

    ref = optimize.least_squares(fun, x0, loss="cauchy", ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=budget)
    assert info["cost"] < 1e-2 * info["cost0"]
    live = np.ones(x0.size, bool)
    live[[0] + [1 + 7 * i for i in range(C - 1)]] = False          # focal entries are dead parameters
    if C == 8:
        # budget-limited: both loops spend the same evaluations and sit at the same point
        assert info["nfev"] == ref.nfev == budget
        np.testing.assert_allclose(info["cost"], ref.cost, rtol=1e-4)
        np.testing.assert_allclose(x_gpu[live], ref.x[live], rtol=1e-5, atol=1e-7)
    if C == 4:
        assert ref.cost < 1e-12 * info["cost0"]
        # converged: the minimum is the generating rig up to the global scale the problem cannot see
        def canon(x):
            p = x[live].reshape(C - 1, 6).copy()
            p[:, 3:] /= np.linalg.norm(p[0, 3:])
            return p
        truth = helpers._ba_x0([{"R": rig["R"][i], "t": rig["t"][i]} for i in range(C)])
        assert info["cost"] < 1e-12 * info["cost0"]
        np.testing.assert_allclose(canon(x_gpu), canon(truth), rtol=1e-5, atol=1e-6)


SOLVED_GOLDENS = ("ba_c3_n24", "ba_c4_n60_solved", "ba_c6_n80_solved", "ba_c8_n100_solved")


def _reference_mode(core, name, mode):
    from mocap_core import helpers, synth
    g = load_golden(name)
    C = g["K"].shape[0]
    helpers.set_core(core)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in g["K"]])
    poses0 = [{"R": g["R_init"][i].copy(), "t": g["t_init"][i].copy()} for i in range(C)]
    with helpers.bundle_adjustment_mode(mode):
        poses, info = helpers.bundle_adjustment(synth.obs_to_reference_array(g["obs"]), poses0, None, return_info=True)
    R = np.array([np.asarray(p["R"], dtype=np.float64) for p in poses])
    t = np.array([np.asarray(p["t"], dtype=np.float64).reshape(3) for p in poses])
    return g, R, t, info


@pytest.mark.parametrize("name", SOLVED_GOLDENS)
def test_reference_mode_scipy_optimizer_reproduces_reference_poses(core, name):
    """helpers.py:287-290 in the reference's own settings (float32 residuals, loss="cauchy", ftol=1e-2), mode
    "scipy": the reference's optimizer call with GPU residual evaluations.  The goldens hold what the reference
    itself returned (R_ba, t_ba, OptimizeResult statistics; oracle/make_golden.py golden_ba).  After the
    float32 cast the GPU residuals are the reference's at every point its optimizer visits, so the whole
    trajectory is: same nfev / njev / status, cost and poses equal (measured: bit-identical; asserted to 1e-9
    to leave room for another libm's last bit in SciPy's own arithmetic)."""
    g, R, t, info = _reference_mode(core, name, "scipy")
    assert [info["nfev"], info["njev"], info["status"]] == g["ba_stats"].tolist()
    np.testing.assert_allclose(info["cost"], g["ba_cost"][0], rtol=1e-9)
    np.testing.assert_allclose(R, g["R_ba"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(t, g["t_ba"], rtol=1e-9, atol=1e-9 * np.abs(g["t_ba"]).max())


@pytest.mark.parametrize("name", SOLVED_GOLDENS)
def test_reference_mode_resident_loop_vs_reference_poses(core, name):
    """The same question for mode "resident" (mocap_ba_solve: float32 data flow restated operation by operation,
    Gram matrix on the matrix cores, trust region on the normal equations instead of SciPy's SVD of J).
    north_star asks for poses within 1e-5 relative.  Yardstick: the reference itself.  The float32 cast of the
    residuals (helpers.py:273) turns a last-bit change of a trial point into 1e-4-relative changes of a few
    Jacobian entries, so the reference's OWN result moves when its start vector is nudged by 1e-15 relative
    (goldens: self_dR / self_dt / self_stats, three draws, produced by the reference run through the harness):
    ~1e-6 at 3 and 4 cameras, ~1e-3 -- with a different evaluation count -- at 8 cameras x 100 points.  A solver
    that cannot share LAPACK's bits lands inside that cloud: asserted here as <= 3 x the reference's own spread
    (and <= 1e-5 where the reference reproduces itself that well)."""
    g, R, t, info = _reference_mode(core, name, "resident")
    d_R = np.abs(R - g["R_ba"]).max()
    d_t = np.abs(t - g["t_ba"]).max() / np.abs(g["t_ba"]).max()
    spread_R, spread_t = float(g["self_dR"].max()), float(g["self_dt"].max())
    print(f"[{name}] resident vs reference: nfev {info['nfev']:.0f}/{g['ba_stats'][0]} njev {info['njev']:.0f}/{g['ba_stats'][1]} "
          f"cost {info['cost']:.12g}/{g['ba_cost'][0]:.12g} dR {d_R:.3g} (reference vs itself {spread_R:.3g}) "
          f"dt_rel {d_t:.3g} ({spread_t:.3g})")
    assert int(info["status"]) == int(g["ba_stats"][2])
    if (g["self_stats"] == g["ba_stats"][None, :]).all():
        # the reference keeps its evaluation counts under the nudge: so does the resident loop
        assert [int(info["nfev"]), int(info["njev"])] == g["ba_stats"][:2].tolist()
        np.testing.assert_allclose(info["cost"], g["ba_cost"][0], rtol=1e-3)
    assert d_R <= max(1e-5, 3 * spread_R) and d_t <= max(1e-5, 3 * spread_t), (d_R, d_t, spread_R, spread_t)
    if spread_R < 3e-6:
        assert d_R < 1e-5 and d_t < 1e-5


def test_residuals_along_the_reference_trajectory(core):
    """Every parameter vector the reference's optimizer evaluated on the 8-camera golden (trial points and
    forward-difference probes, ba_eval_xs): the GPU residuals against the C oracle's, before and after the
    float32 cast of helpers.py:273."""
    from oracle import c_oracle
    g = load_golden("ba_c8_n100_solved")
    core.set_cameras(g["K"], g["R_init"], g["t_init"])
    xs = g["ba_eval_xs"]
    r = core.ba_residuals(xs, g["obs"])
    ref = c_oracle.COracle(g["K"], g["R_init"], g["t_init"]).ba_residuals(xs, g["obs"])
    assert np.array_equal(np.isnan(r), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(r[ok], ref[ok], rtol=1e-9)                     # measured: 7e-16
    assert (r[ok].astype(np.float32) == ref[ok].astype(np.float32)).mean() > 0.9999   # measured: all equal


def test_helpers_bundle_adjustment_api(core):
    """Drop-in call shape of helpers.bundle_adjustment (helpers.py:244): list of {"R","t"} back, one
    socket emit, camera 0 pinned at identity."""
    from mocap_core import helpers, synth
    helpers.set_core(core)
    rig = synth.ring_rig(4)
    rng = np.random.default_rng(70)
    obs, _ = synth.make_ba_observations(rig, 120, seed=70)
    init = synth.perturb_rig(rig, rng)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    poses0 = [{"R": init["R"][i], "t": init["t"][i].reshape(3, 1)} for i in range(4)]

    class Sock:
        n = 0

        def emit(self, name, payload):
            assert name == "camera-pose" and len(payload["camera_poses"]) == 4
            Sock.n += 1
    for mode in ("resident", "scipy"):
        with helpers.bundle_adjustment_mode(mode):
            out = helpers.bundle_adjustment(synth.obs_to_reference_array(obs), poses0, Sock())
        assert len(out) == 4 and np.array_equal(out[0]["R"], np.eye(3))
        assert all(np.asarray(p["R"]).shape == (3, 3) and np.asarray(p["t"]).size == 3 for p in out)
    assert helpers._state["ba_mode"] == helpers.DEFAULT_BA_MODE
    assert Sock.n >= 2            # progress events: tests/test_gpu_boundary.py counts them


# ---------------------------------------------------------------------------------------------------------------
# The launch-ahead hand-over (a kernel resident on the GPU polling a host mailbox for its base point) under a host
# that is late: device watchdog -> abandoned launch -> the host notices and repeats the linearisation.
def _ba_case(n_pts=400):
    from mocap_core import helpers, synth
    rig = synth.ring_rig(8)
    rng = np.random.default_rng(77)
    obs, _ = synth.make_ba_observations(rig, n_pts, seed=77)
    init = synth.perturb_rig(rig, rng)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
    return rig, init, obs, x0


_ABANDON_CHILD = r"""
import json, sys, time
import numpy as np
sys.path[:0] = [{root!r}, {pkg!r}]
sys.path.insert(0, {tests!r})
import torch
from mocap_core import capi
from test_gpu_ba import _ba_case
rig, init, obs, x0 = _ba_case()
core = capi.MocapCore(0)
core.set_cameras(rig["K"], init["R"], init["t"])
t0 = time.perf_counter()
x, info = core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=30)
print(json.dumps({{"x": x.tolist(), "info": info, "seconds": time.perf_counter() - t0}}))
"""


def _run_child(env_extra):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _ABANDON_CHILD.format(root=root, pkg=os.path.join(root, "low-cost-mocap_amd"), tests=os.path.join(root, "tests"))
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_abandoned_launch_ahead_is_detected_and_repeated():
    """MOCAP_BA_DEBUG_HOST_STALL holds the host up for 2.6 s before its 6th hand-over: the resident kernel's 2 s device
    watchdog fires, it writes the abandon mark, the host repeats that linearisation with the point passed by value and
    carries on -- same iterates as the undisturbed solve, bit for bit (the arithmetic does not depend on how the base
    point reached the kernel), one relaunch counted, and nowhere near the 20 s give-up."""
    plain = _run_child({})
    late = _run_child({"MOCAP_BA_DEBUG_HOST_STALL": "5:2600"})
    assert plain["info"]["relaunches"] == 0
    assert late["info"]["relaunches"] >= 1
    assert late["seconds"] < 12.0
    assert late["info"]["nfev"] == plain["info"]["nfev"] and late["info"]["iterations"] == plain["info"]["iterations"]
    assert np.array_equal(np.array(late["x"]), np.array(plain["x"]))


def test_slow_progress_callback_does_not_strand_a_resident_kernel(core):
    """The reference emits a socket event per evaluation (helpers.py:274); the callback is foreign code that may block
    (GIL, an eventlet yield).  It runs with nothing parked on the GPU: a 2.3 s callback -- longer than the device
    watchdog -- changes neither the result nor the relaunch count."""
    import time
    rig, init, obs, x0 = _ba_case()
    core.set_cameras(rig["K"], init["R"], init["t"])
    x_ref, info_ref = core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=12)
    calls = []

    def cb(x):
        calls.append(x.copy())
        if len(calls) == 2:
            time.sleep(2.3)
    core.set_ba_progress(cb)
    try:
        x, info = core.ba_solve(x0, obs, ftol=0.0, xtol=0.0, gtol=0.0, max_iter=12)
    finally:
        core.set_ba_progress(None)
    assert len(calls) >= 2 and info["relaunches"] == 0
    assert np.array_equal(x, x_ref) and info["nfev"] == info_ref["nfev"]


def test_no_speculation_mode_with_launch_ahead_default():
    """MOCAP_BA_NO_SPECULATION=1 (cost-only trial evaluations, the documented A/B mode) used to queue its kernels
    behind a resident launch-ahead kernel that was waiting for a mailbox write: 2 s per evaluation, then a failure.
    It now switches the launch-ahead off for the solve."""
    a = _run_child({"MOCAP_BA_NO_SPECULATION": "1"})
    b = _run_child({})
    assert a["seconds"] < 8.0 and a["info"]["relaunches"] == 0
    np.testing.assert_allclose(np.array(a["x"]), np.array(b["x"]), rtol=0, atol=1e-9)


def test_ba_solve_keeps_its_8_double_info_abi(core):
    """mocap_ba_solve writes exactly 8 doubles (the buffer size it was first exported with); the longer record is
    mocap_ba_solve_ex's, bounded by the length the caller states."""
    import ctypes
    rig, init, obs, x0 = _ba_case(120)
    core.set_cameras(rig["K"], init["R"], init["t"])
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    for fn, extra, n_written in ((core.lib.mocap_ba_solve, (), 8), (core.lib.mocap_ba_solve_ex, (9,), 9)):
        x = np.array(x0, dtype=np.float64)
        info = np.full(12, -7.0)
        rc = fn(core._h, x.ctypes.data_as(ctypes.c_void_p), obs.shape[0], obs.ctypes.data_as(ctypes.c_void_p), 1e-2, 1e-8,
                1e-8, 0, 1, 1, info.ctypes.data_as(ctypes.c_void_p), *extra)
        assert rc in (0, -5)
        assert np.all(info[n_written:] == -7.0) and info[0] >= 1 and info[6] > 0


# ----------------------------------------------------------------------------- BASELINE configs[3]'s own size
# "Bundle-adjustment calibration: 8 cams, 2000 frames x 8 markers": 16 000 points.  The launch takes another grid and a
# deeper reduction tree than anything at 1 000 points (250 chunks of 64 points, two levels of the 16-ary last-arriver
# tree; no launch-ahead: N > 2048), so the three rungs of the ladder are repeated at that size.
N_CALIB = 16_000


@pytest.fixture(scope="module")
def calib16k():
    from mocap_core import helpers, synth
    rig = synth.ring_rig(8)
    rng = np.random.default_rng(81)
    obs, _ = synth.make_ba_observations(rig, N_CALIB, seed=81, dropout=0.05)
    init = synth.perturb_rig(rig, rng)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
    return rig, init, obs, x0


def test_residuals_vs_c_oracle_16k_points(core, calib16k):
    from oracle import c_oracle
    rig, init, obs, x0 = calib16k
    rng = np.random.default_rng(82)
    core.set_cameras(rig["K"], init["R"], init["t"])
    xs = np.stack([x0, x0 + rng.normal(0, 1e-3, x0.size)])
    r = core.ba_residuals(xs, obs)
    ref = c_oracle.COracle(rig["K"], init["R"], init["t"]).ba_residuals(xs, obs)
    assert r.shape == (2, N_CALIB) and np.array_equal(np.isnan(r), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert ok.sum() > 2 * 0.99 * N_CALIB
    np.testing.assert_allclose(r[ok], ref[ok], rtol=1e-3, atol=1e-12)   # float32-boundary flips of cv.projectPoints' rounding
    assert np.median(np.abs(r[ok] - ref[ok]) / ref[ok]) < 1e-9
    assert (np.abs(r[ok] - ref[ok]) <= 1e-9 * ref[ok]).mean() > 0.99


def test_gram_mfma_vs_numpy_16k_points(core, calib16k):
    """J^T J / J^T f of the one-launch linearisation at 16 000 points (250 chunks, two-level reduction tree) against NumPy on
    the J the same launch returned, in both residual precisions; J itself against SciPy's own float32 differencing."""
    from scipy.optimize._lsq.common import scale_for_robust_loss_function
    from scipy.optimize._lsq.least_squares import construct_loss_function
    from scipy.optimize._numdiff import approx_derivative
    rig, init, obs, x0 = calib16k
    core.set_cameras(rig["K"], init["R"], init["t"])
    dead = [0] + [1 + 7 * i for i in range(7)]
    for f32 in (False, True):
        ne = core.ba_normal_eq(x0, obs, f32_residuals=f32, use_cauchy=True, want_J=True)
        J = ne["J"]
        assert J.shape == (ne["m"], 50) and ne["m"] > 0.99 * N_CALIB
        G = J.T @ J
        np.testing.assert_allclose(ne["JtJ"], G, rtol=1e-11, atol=1e-12 * np.abs(G).max())
        assert not J[:, dead].any() and not ne["JtJ"][dead].any()

    def fun(x):                                                   # = the reference's residual_function
        r = core.ba_residuals(x, obs)[0]
        return r[~np.isnan(r)].astype(np.float32)

    f0 = fun(x0)
    J_ref = approx_derivative(fun, x0, method="2-point", f0=f0)
    rho = construct_loss_function(f0.size, "cauchy", 1.0)(f0)
    J_ref, f_ref = scale_for_robust_loss_function(J_ref, f0.copy(), rho)
    assert ne["m"] == f0.size
    np.testing.assert_allclose(ne["J"], J_ref, rtol=1e-13, atol=1e-13 * np.abs(J_ref).max())
    np.testing.assert_allclose(ne["cost"], 0.5 * np.sum(rho[0]), rtol=2e-7)
    g_ref = J_ref.T.dot(f_ref)
    np.testing.assert_allclose(ne["Jtr"], g_ref, rtol=1e-10, atol=1e-11 * np.abs(g_ref).max())


def test_trust_region_step_vs_scipy_16k_points(core, calib16k):
    """One trust-region step on the 16 000-point normal equations against scipy's solve_lsq_trust_region on the SVD of the
    same J (scipy _lsq/trf.py:448,495)."""
    from scipy.linalg import svd
    from scipy.optimize._lsq.common import solve_lsq_trust_region
    rig, init, obs, x0 = calib16k
    core.set_cameras(rig["K"], init["R"], init["t"])
    ne = core.ba_normal_eq(x0, obs, f32_residuals=False, use_cauchy=True, want_J=True)
    r0 = core.ba_residuals(x0, obs)[0]
    f = r0[~np.isnan(r0)]
    z = f * f
    f_scaled = f * (1 / (1 + z)) / np.sqrt(np.maximum(1 / (1 + z) - 2 * z / (1 + z) ** 2, np.finfo(float).eps))
    J = ne["J"]
    m, n = J.shape
    U, s, Vt = svd(J, full_matrices=False)
    uf = U.T @ f_scaled
    dead = [0] + [1 + 7 * i for i in range(7)]
    live = np.setdiff1d(np.arange(n), dead)
    np.testing.assert_allclose(ne["Jtr"], J.T @ f_scaled, rtol=1e-9, atol=1e-9 * np.abs(ne["Jtr"]).max())
    for Delta in (np.linalg.norm(x0), 1e-2, 1e-4):
        p_ref, a_ref, _ = solve_lsq_trust_region(n, m, uf, s, Vt.T, Delta, initial_alpha=0.0)
        p, a, info = core.ba_trust_region_step(ne["JtJ"], ne["Jtr"], m, Delta, alpha=0.0, method=0)
        assert info["live"] == n - len(dead) and not p[dead].any()
        np.testing.assert_allclose(a, a_ref, rtol=1e-8)
        np.testing.assert_allclose(p[live], p_ref[live], rtol=1e-7, atol=1e-8 * np.abs(p_ref).max())


def test_resident_solve_16k_points_reduces_the_cost_like_scipy(core, calib16k):
    """The whole loop at the calibration's own size: mocap_ba_solve and scipy.optimize.least_squares on GPU residuals
    (reference settings) both stop on ftol = 1e-2, at costs within that tolerance's reach of each other."""
    from scipy import optimize
    rig, init, obs, x0 = calib16k
    core.set_cameras(rig["K"], init["R"], init["t"])
    x, info = core.ba_solve(x0, obs, ftol=1e-2, f32_residuals=True, use_cauchy=True)

    def fun(p):
        r = core.ba_residuals(p, obs)[0]
        return r[~np.isnan(r)].astype(np.float32)

    ref = optimize.least_squares(fun, x0, loss="cauchy", ftol=1e-2, max_nfev=12)
    assert int(info["m"]) == fun(x0).size and info["cost"] < 0.5 * info["cost0"]
    if ref.status > 0:
        np.testing.assert_allclose(info["cost"], ref.cost, rtol=0.1)


# ----------------------------------------------------------------------------- default mode: one call per Jacobian
def test_batched_residuals_equal_single_calls_bit_for_bit(core):
    """mocap_ba_residuals over P parameter vectors at once = P single calls, to the bit (the batch is what the default
    mode's Jacobian rides on; grid.y = parameter vector, same kernel, same per-point arithmetic)."""
    from mocap_core import helpers, synth
    rig = synth.ring_rig(8)
    rng = np.random.default_rng(91)
    obs, _ = synth.make_ba_observations(rig, 1000, seed=91, dropout=0.08)
    init = synth.perturb_rig(rig, rng)
    core.set_cameras(rig["K"], init["R"], init["t"])
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    x0 = helpers._ba_x0([{"R": init["R"][i], "t": init["t"][i]} for i in range(8)])
    X = np.stack([x0 + rng.normal(0, 3e-4, x0.size) * (rng.random(x0.size) < 0.1) for _ in range(51)])
    batch = core.ba_residuals(X, obs)
    for p in range(X.shape[0]):
        single = core.ba_residuals(X[p], obs)[0]
        assert np.array_equal(batch[p], single, equal_nan=True), p


@pytest.mark.parametrize("name", SOLVED_GOLDENS)
def test_default_mode_batched_jacobian_is_scipys_own(core, name, monkeypatch):
    """helpers.bundle_adjustment (mode "scipy") with its one-call Jacobian against the same call with SciPy differencing
    by itself (jac="2-point": n + 1 host round trips per Jacobian, MOCAP_BA_BATCHED_JAC=0): identical OptimizeResult
    statistics and identical poses -- bit for bit, not to a tolerance -- and n + 1 `camera-pose` events per Jacobian plus
    one per trial point, as the reference emits (helpers.py:274)."""
    class Sock:
        def __init__(self):
            self.n = 0

        def emit(self, event, payload):
            assert event == "camera-pose"
            self.n += 1

    from mocap_core import helpers, synth
    g = load_golden(name)
    C = g["K"].shape[0]
    helpers.set_core(core)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in g["K"]])
    out = {}
    for batched in ("1", "0"):
        monkeypatch.setenv("MOCAP_BA_BATCHED_JAC", batched)
        poses0 = [{"R": g["R_init"][i].copy(), "t": g["t_init"][i].copy()} for i in range(C)]
        sock = Sock()
        with helpers.bundle_adjustment_mode("scipy"):
            poses, info = helpers.bundle_adjustment(synth.obs_to_reference_array(g["obs"]), poses0, sock, return_info=True)
        out[batched] = (np.array([np.asarray(p["R"], dtype=np.float64) for p in poses]),
                        np.array([np.asarray(p["t"], dtype=np.float64).reshape(3) for p in poses]), info, sock.n)
    (R1, t1, i1, e1), (R0, t0, i0, e0) = out["1"], out["0"]
    assert [i1["nfev"], i1["njev"], i1["status"]] == [i0["nfev"], i0["njev"], i0["status"]] == g["ba_stats"].tolist()
    assert i1["cost"] == i0["cost"] and np.array_equal(R1, R0) and np.array_equal(t1, t0)
    n = 1 + 7 * (C - 1)
    assert e1 == e0 == i1["nfev"] + n * i1["njev"] + 1          # + the final poses (index.py:277)


@pytest.mark.gpu
def test_device_subproblem_microbench_solves_the_system(core):
    """tests/native/libmocap_trbench.so (test-only, NOT in the product library: the measurement aid behind DESIGN 3.4's "the
    subproblem stays on the host", profiles/r05_ba_device_subproblem.txt): the one-wave Cholesky + triangular solves it times
    must actually solve (B + a I) p = -g -- against plain double loops on the host."""
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "native", "libmocap_trbench.so")
    assert os.path.exists(path), "build it with `make -C tests/native` (__graft_entry__.build does)"
    lib = ctypes.CDLL(path)
    lib.trbench_run.restype = ctypes.c_int
    lib.trbench_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = np.zeros(3)
    assert core is not None and lib.trbench_run(0, 8, out.ctypes.data) == 0
    assert out[2] < 1e-11                       # max |p - p_host| / max |p_host|
    assert 0.0 < out[0] < out[1] < 1e4          # us per factorisation < us per shift
