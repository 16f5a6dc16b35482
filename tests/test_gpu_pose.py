"""GPU parity of the initial pose estimation (calculate_camera_pose, reference
index.py:229-270) against the oracle and the reference-run goldens.

Tolerances: the same RANSAC sample must win (inlier counts and iteration numbers are integers: exact);
the fundamental matrix of that sample and the poses agree to 1e-6 / 1e-5 relative -- the 7-point null
space is found by elimination in the core and by LAPACK's SVD in the oracle, two bases of the same space."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from mocap_core import synth
from oracle import cv_pose_restate as cp
from oracle import pose_oracle

pytestmark = pytest.mark.gpu


def _pairs(obs, ci):
    a, b = obs[:, ci], obs[:, ci + 1]
    ok = ~(np.isnan(a).any(axis=1) | np.isnan(b).any(axis=1))
    return a[ok].astype(np.float32), b[ok].astype(np.float32)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_find_fundamental_replays_opencv_ransac(core, seed):
    rig = synth.ring_rig(3)
    obs, _ = synth.make_ba_observations(rig, 150 + 40 * seed, seed=20 + seed, noise_px=0.3 + 0.2 * seed)
    obs = np.trunc(obs)
    p1, p2 = _pairs(obs, seed % 2)
    F, mask, info = core.find_fundamental(p1, p2, 1.0, 0.99999)
    Fr, maskr, infor = cp.find_fundamental_mat(p1, p2, cp.FM_RANSAC, 1.0, 0.99999, return_info=True)
    assert info == infor                                             # inliers, iterations, winning iteration
    assert np.array_equal(mask, maskr.ravel())
    np.testing.assert_allclose(F, Fr, rtol=1e-6, atol=1e-9 * np.abs(Fr).max())
    # and it is a fundamental matrix of the rig: true correspondences satisfy it to about a pixel
    h1, h2 = np.c_[p1, np.ones(len(p1))], np.c_[p2, np.ones(len(p2))]
    lines = h1 @ F.T
    d = np.abs(np.einsum("ni,ni->n", h2, lines)) / np.hypot(lines[:, 0], lines[:, 1])
    assert np.median(d) < 1.0


def test_find_fundamental_long_run_many_batches(core):
    """30 % gross outliers: the loop needs hundreds of iterations, i.e. several GPU batches, and the
    replayed bookkeeping still lands on the sequential loop's model."""
    rng = np.random.default_rng(5)
    rig = synth.ring_rig(2)
    obs, _ = synth.make_ba_observations(rig, 300, seed=30, dropout=0.0)
    obs = np.trunc(obs)
    bad = rng.random(300) < 0.3
    obs[bad, 1] = rng.uniform(0, 320, (bad.sum(), 2)).astype(int)
    p1, p2 = _pairs(obs, 0)
    F, mask, info = core.find_fundamental(p1, p2, 1.0, 0.99999)
    Fr, maskr, infor = cp.find_fundamental_mat(p1, p2, cp.FM_RANSAC, 1.0, 0.99999, return_info=True)
    assert info == infor and info["iterations"] > 128
    assert np.array_equal(mask, maskr.ravel())
    np.testing.assert_allclose(F, Fr, rtol=1e-6, atol=1e-9 * np.abs(Fr).max())


@pytest.mark.parametrize("name", golden_names("pose_"))
def test_initial_poses_match_reference_golden(core, name):
    g = load_golden(name)
    R, t, info = core.initial_poses(g["obs"], g["K"])
    np.testing.assert_allclose(R, g["ref_R"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(t, g["ref_t"], rtol=1e-5, atol=1e-5)
    assert (info[:, 1] >= 7).all()


def test_initial_poses_vs_oracle_and_truth(core):
    C = 5
    rig = synth.ring_rig(C)
    obs, _ = synth.make_ba_observations(rig, 500, seed=41, dropout=0.1)
    obs = np.trunc(obs)
    R, t, info = core.initial_poses(obs, rig["K"])
    Ro, to, infos = pose_oracle.initial_poses(obs, [k for k in rig["K"]], return_info=True)
    np.testing.assert_allclose(R, Ro, atol=1e-5)
    np.testing.assert_allclose(t, to, rtol=1e-5, atol=1e-5)
    assert [int(i[0]) for i in info] == [d["n"] for d in infos]
    assert [int(i[1]) for i in info] == [d["inliers"] for d in infos]
    assert [int(i[2]) for i in info] == [d["iterations"] for d in infos]
    # the first pair is a proper relative pose: direction of t and rotation close to the rig's
    tt = rig["t"][1] / np.linalg.norm(rig["t"][1])
    assert np.abs(R[1] - rig["R"][1]).max() < 0.08
    assert np.linalg.norm(t[1] / np.linalg.norm(t[1]) - tt) < 0.08


def test_handler_mirror_end_to_end(core):
    """helpers.calculate_camera_pose = index.py:229-281: initial poses -> bundle adjustment -> error."""
    from mocap_core import helpers
    from oracle.ref_harness import NullSocket
    C = 4
    rig = synth.ring_rig(C)
    obs, _ = synth.make_ba_observations(rig, 300, seed=42)
    obs = np.trunc(obs)
    helpers.set_core(core)
    helpers.set_camera_params([{"intrinsic_matrix": k.tolist()} for k in rig["K"]])
    sock = NullSocket()
    poses, err = helpers.calculate_camera_pose({"cameraPoints": synth.obs_to_reference_array(obs, as_int=True).tolist()}, sock)
    assert len(poses) == C and sock.count >= 1 and np.isfinite(err)
    assert err < 50.0                                 # px^2, mean squared reprojection error after BA


def test_too_few_points_is_an_error_not_a_guess(core):
    from mocap_core.capi import MocapError
    rig = synth.ring_rig(2)
    obs, _ = synth.make_ba_observations(rig, 10, seed=1, dropout=0.0)
    with pytest.raises(MocapError):
        core.initial_poses(obs, rig["K"])
