"""TEST INFRASTRUCTURE ONLY (oracle/).  Pin-when-possible for the OpenCV arithmetic the reference calls.

The reference's hot path goes through OpenCV (un-vendored, un-versioned: SURVEY.md section 8c) at
    helpers.py:362      cv.sfm.fundamentalFromProjections
    helpers.py:363      cv.computeCorrespondEpilines
    helpers.py:231-237  cv.projectPoints
and the rows either side of it at
    helpers.py:73-82, 145-152   cv.undistort / GaussianBlur / filter2D / cvtColor / threshold / findContours / moments
    index.py:246-248            cv.findFundamentalMat(FM_RANSAC) / cv.sfm.essentialFromFundamental / motionFromEssential
This image has no cv2, so oracle/cv_restate.py, cv_image_restate.py and cv_pose_restate.py restate those functions
from the published OpenCV sources and the goldens are "parity unpinned" there.  On ANY machine where `import cv2`
works, this script calls the REAL functions on seeded inputs and writes inputs + outputs to tests/golden/cv2_*.npz;
tests/test_oracle_cv2_pin.py then compares the restatements with them (it skips while the files are absent).

    python -m oracle.make_cv2_golden          # or: python -m oracle.make_golden --with-cv2

Functions of opencv_contrib's sfm module are pinned when the build has them (`cv2.sfm`), skipped otherwise.
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "low-cost-mocap_amd"))
OUT = os.path.join(_ROOT, "tests", "golden")


def real_cv2():
    """The real OpenCV module, or None (also None for the stub namespace of oracle/ref_harness.py)."""
    try:
        import cv2
    except Exception:
        return None
    return cv2 if hasattr(cv2, "__version__") and hasattr(cv2, "getBuildInformation") else None


def inputs_hot_path(seed=0):
    from mocap_core import synth
    rng = np.random.default_rng(seed)
    rig = synth.ring_rig(8)
    P = np.array([rig["K"][i] @ np.c_[rig["R"][i], rig["t"][i]] for i in range(8)])
    pts = rng.uniform(0, 320, (64, 2)).astype(np.float32)
    X = rng.uniform(-0.7, 0.7, (64, 3)) + np.array([0, 0, 3.0])
    return rig, P, pts, X


def inputs_image(seed=1):
    from mocap_core import synth
    rig = synth.ring_rig(1)
    images, _ = synth.render_camera_frames(rig, 1, 10, seed=seed, noise_levels=3)
    return images[0, 0], np.array(rig["K"][0]), np.array(synth.REFERENCE_DISTORTION, dtype=np.float64)


def inputs_pose(seed=2):
    from mocap_core import synth
    rig = synth.ring_rig(2)
    obs, _ = synth.make_ba_observations(rig, 200, seed=seed, dropout=0.0, noise_px=0.3)
    return np.trunc(obs[:, 0]).astype(np.float32), np.trunc(obs[:, 1]).astype(np.float32), rig


def main():
    cv2 = real_cv2()
    if cv2 is None:
        print("no real cv2 importable here: nothing written (tests/test_oracle_cv2_pin.py keeps skipping)")
        return 1
    os.makedirs(OUT, exist_ok=True)
    ver = np.array(cv2.__version__)
    has_sfm = hasattr(cv2, "sfm")
    # ---- hot path
    rig, P, pts, X = inputs_hot_path()
    out = dict(version=ver, P=P, pts=pts, X=X, K=rig["K"], R=rig["R"], t=rig["t"])
    if has_sfm:
        F = np.array([[cv2.sfm.fundamentalFromProjections(P[a], P[b]) for b in range(8)] for a in range(8)])
        out["F"] = F
    else:
        from oracle import cv_restate
        F = np.array([[cv_restate.fundamental_from_projections(P[a], P[b]) for b in range(8)] for a in range(8)])
    out["F_used"] = F
    out["epilines"] = np.array([cv2.computeCorrespondEpilines(pts.reshape(-1, 1, 2), 1, F[0, b]).reshape(-1, 3)
                                for b in range(1, 8)])
    out["projected"] = np.array([cv2.projectPoints(np.expand_dims(X, 0).astype(np.float32), rig["R"][c], rig["t"][c],
                                                   rig["K"][c], np.array([]))[0].reshape(-1, 2) for c in range(8)])
    np.savez_compressed(os.path.join(OUT, "cv2_hot_path.npz"), **out)
    # ---- image stage
    raw, K, dist = inputs_image()
    img = np.ascontiguousarray(raw)
    und = cv2.undistort(img, K, dist)
    blur = cv2.GaussianBlur(und, (9, 9), 0)
    kernel = np.array([[-2, -1, -1, -1, -2], [-1, 1, 3, 1, -1], [-1, 3, 4, 3, -1], [-1, 1, 3, 1, -1], [-2, -1, -1, -1, -2]])
    filt = cv2.filter2D(blur, -1, kernel)
    bgr = cv2.cvtColor(filt, cv2.COLOR_RGB2BGR)
    grey = cv2.cvtColor(bgr, cv2.COLOR_RGB2GRAY)
    mask = cv2.threshold(grey, 255 * 0.2, 255, cv2.THRESH_BINARY)[1]
    contours, hierarchy = cv2.findContours(mask, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)
    m = np.array([[cv2.moments(c)[k] for k in ("m00", "m10", "m01")] for c in contours]).reshape(-1, 3)
    np.savez_compressed(os.path.join(OUT, "cv2_image_stage.npz"), version=ver, raw=img, K=K, dist=dist, kernel=kernel,
                        undistorted=und, blurred=blur, filtered=filt, grey=grey, mask=mask,
                        n_contours=np.array(len(contours)), contour_first_points=np.array([c[0, 0] for c in contours]).reshape(-1, 2),
                        contour_lengths=np.array([len(c) for c in contours]),
                        hierarchy=np.array(hierarchy).reshape(-1, 4) if hierarchy is not None else np.zeros((0, 4), int),
                        moments=m)
    # ---- initial poses
    p1, p2, rig2 = inputs_pose()
    Fm, inl = cv2.findFundamentalMat(p1, p2, cv2.FM_RANSAC, 1, 0.99999)
    out = dict(version=ver, p1=p1, p2=p2, K=rig2["K"], F=Fm, inliers=inl.reshape(-1))
    if has_sfm:
        E = cv2.sfm.essentialFromFundamental(Fm, rig2["K"][0], rig2["K"][1])
        Rs, ts = cv2.sfm.motionFromEssential(E)
        out.update(E=E, Rs=np.array(Rs), ts=np.array(ts).reshape(-1, 3))
    np.savez_compressed(os.path.join(OUT, "cv2_pose_init.npz"), **out)
    print("wrote cv2_hot_path / cv2_image_stage / cv2_pose_init (.npz), OpenCV", cv2.__version__, "sfm:", has_sfm)
    return 0


if __name__ == "__main__":
    sys.exit(main())
