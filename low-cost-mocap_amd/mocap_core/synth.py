"""Seeded synthetic rigs and blob streams (SURVEY.md section 8d / BASELINE.md section 3).

The reference ships no data, so every workload is generated: C cameras on a ring
looking at the origin, re-expressed relative to camera 0 (the reference fixes
camera 0 at (I, 0): computer_code/api/index.py:235-238, helpers.py:250-253),
markers uniform in a cube, pinhole projection + Gaussian pixel noise, `int()`
truncation exactly like the reference blob finder (helpers.py:154-155), a random
per-camera permutation and per-(camera, marker) dropout.

Layouts produced here are the ones the C-ABI consumes (include/mocap_core.h):
  blobs  f32 [F][C][M_max][2]   (unused slots = NaN)
  counts i32 [F][C]
"""
import numpy as np

DEFAULT_K = [[320.0, 0.0, 160.0], [0.0, 320.0, 160.0], [0.0, 0.0, 1.0]]  # camera-params.json:3-5
VGA_K = [[640.0, 0.0, 320.0], [0.0, 640.0, 240.0], [0.0, 0.0, 1.0]]
# Stress config (BASELINE.json configs[4], 64 cams x 256 markers): the reference enumerates the full
# Cartesian product of gated hits over all cameras (helpers.py:394-400), so with 63 other cameras
# even 0.1 false hits per camera explodes.  The virtual sensor is therefore 16 k x 16 k px with
# sub-pixel (float) centroids and the gate is STRESS_GATE_PX: ~0.02 false hits per (root, camera).
STRESS_K = [[12000.0, 0.0, 8000.0], [0.0, 12000.0, 8000.0], [0.0, 0.0, 1.0]]
STRESS_GATE_PX = 0.5


def stress_rig(num_cameras=64):
    return ring_rig(num_cameras, K=STRESS_K, image_size=(16000, 16000))


def make_stress_stream(rig, n_frames, n_markers=256, seed=0):
    """64 x 256-style stream: float centroids, 0.02 px noise, markers in a +-1.5 m cube."""
    return make_blob_stream(rig, n_frames, n_markers, seed=seed, noise_px=0.02, dropout=0.05,
                            half_extent=1.5, min_sep=0.05, truncate=False)


def make_stress_stream_chunked(rig, n_frames, n_markers=256, seed=0, chunk=256, threads=None):
    """The same distribution as make_stress_stream, generated in independent chunks of `chunk` frames (chunk k: seed
    1 000 003 * (seed + 1) + k) on a thread pool -- NumPy releases the GIL in the generator's heavy calls, so the host time of
    a 12 500-frame stress batch drops from minutes to seconds on the bench box's cores.  The result depends on (seed, chunk)
    only, never on the thread count.  bench.py's 64 x 256 streams come from here since round 6."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n_frames = int(n_frames)
    sizes = [min(chunk, n_frames - lo) for lo in range(0, n_frames, chunk)]
    threads = threads or min(len(sizes), os.cpu_count() or 1, 64)

    def gen(k):
        return make_stress_stream(rig, sizes[k], n_markers, seed=1_000_003 * (int(seed) + 1) + k)
    with ThreadPoolExecutor(max(1, threads)) as ex:
        parts = list(ex.map(gen, range(len(sizes))))
    blobs = np.concatenate([p[0] for p in parts], axis=0)
    counts = np.concatenate([p[1] for p in parts], axis=0)
    truth = {"points_cam0": np.concatenate([p[2]["points_cam0"] for p in parts], axis=0),
             "ident": np.concatenate([p[2]["ident"] for p in parts], axis=0)}
    return blobs, counts, truth


def _look_at(cam_pos, target=np.zeros(3)):
    """World->camera rotation with +z along the optical axis, +y roughly world -z (image down)."""
    z = target - cam_pos
    z = z / np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z])


def ring_rig(num_cameras, radius=3.0, height=1.5, K=None, image_size=None):
    """Returns dict(K (C,3,3), R (C,3,3), t (C,3), image_size (w,h)) relative to camera 0.

    64-camera stress rig: two rings of C/2 at heights 1.0 / 2.5 m."""
    C = int(num_cameras)
    K = np.array(DEFAULT_K if K is None else K, dtype=np.float64)
    if image_size is None:
        image_size = (int(round(2 * K[0, 2])), int(round(2 * K[1, 2])))
    Rw, tw = [], []
    for i in range(C):
        if C >= 32:
            half = C // 2
            ring, k, n = (0, i, half) if i < half else (1, i - half, C - half)
            h = 1.0 if ring == 0 else 2.5
            ang = 2 * np.pi * (k + 0.5 * ring) / n
        else:
            h = height
            ang = 2 * np.pi * i / C
        pos = np.array([radius * np.cos(ang), radius * np.sin(ang), h])
        R = _look_at(pos)
        Rw.append(R)
        tw.append(-R @ pos)
    Rw, tw = np.array(Rw), np.array(tw)
    R0, t0 = Rw[0], tw[0]
    R = np.array([Rw[i] @ R0.T for i in range(C)])
    t = np.array([tw[i] - R[i] @ t0 for i in range(C)])
    R[0] = np.eye(3)
    t[0] = 0.0
    # markers live around the world origin == this point in camera-0 coordinates
    centre = R0 @ np.zeros(3) + t0
    return {"K": np.repeat(K[None], C, axis=0), "R": R, "t": t,
            "image_size": image_size, "centre": centre, "R0": R0}


def rig_to_pose_dicts(rig):
    """Poses in the JSON shape the reference's socket API carries ({"R": 3x3, "t": 3})."""
    return [{"R": rig["R"][i].tolist(), "t": rig["t"][i].tolist()} for i in range(len(rig["R"]))]


def perturb_rig(rig, rng, rot_sigma=0.02, trans_sigma=0.05):
    """Initial poses for BA: truth (+) N(0, rot_sigma rad) / N(0, trans_sigma m); camera 0 untouched."""
    from scipy.spatial.transform import Rotation
    out = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in rig.items()}
    for i in range(1, len(rig["R"])):
        dR = Rotation.from_rotvec(rng.normal(0, rot_sigma, 3)).as_matrix()
        out["R"][i] = dR @ rig["R"][i]
        out["t"][i] = rig["t"][i] + rng.normal(0, trans_sigma, 3)
    return out


def _sample_markers(rng, n_frames, n_markers, half_extent, min_sep):
    """Markers uniform in the cube; the higher-indexed member of every pair closer than min_sep is drawn again (up to 20
    rounds).  Distances are taken in chunks of frames, and from the second round on only for the frames that were changed:
    the same `bad` masks, hence the same draws and the same streams as the one-shot [F][M][M] form this replaces (which
    needed 1.6 GB per round at 1 024 frames of 256 markers: 48 s; now 2 s)."""
    pts = rng.uniform(-half_extent, half_extent, size=(n_frames, n_markers, 3))
    if min_sep > 0 and n_markers > 1:
        upper = np.triu(np.ones((n_markers, n_markers), bool), 1)[None]
        ar = np.arange(n_markers)
        chunk = max(1, (1 << 21) // (n_markers * n_markers))
        active = np.arange(n_frames)
        for _ in range(20):
            bad = np.zeros((n_frames, n_markers), bool)
            for lo in range(0, active.size, chunk):
                idx = active[lo:lo + chunk]
                p = pts[idx]
                d = np.linalg.norm(p[:, :, None, :] - p[:, None, :, :], axis=-1)
                d[:, ar, ar] = np.inf
                # resample the higher-indexed member of every too-close pair
                bad[idx] = (upper & (d < min_sep)).any(axis=1)
            if not bad.any():
                break
            pts[bad] = rng.uniform(-half_extent, half_extent, size=(int(bad.sum()), 3))
            active = np.nonzero(bad.any(axis=1))[0]
    return pts


def make_blob_stream(rig, n_frames, n_markers, seed=0, noise_px=0.3, dropout=0.05,
                     half_extent=0.7, min_sep=0.05, truncate=True, m_max=None, shuffle=True, world=None):
    """Returns (blobs f32 [F][C][M_max][2] NaN-padded, counts i32 [F][C], truth dict).
    world: None (markers sampled in the cube), or a function (sampled [F][M][3]) -> [F][M][3] that edits them (tests)."""
    rng = np.random.default_rng(seed)
    C = len(rig["R"])
    F, M = int(n_frames), int(n_markers)
    m_max = M if m_max is None else int(m_max)
    sampled = _sample_markers(rng, F, M, half_extent, min_sep)
    world = sampled if world is None else np.asarray(world(sampled), dtype=np.float64)
    # world -> camera-0 coordinates
    X0 = world @ rig["R0"].T + rig["centre"]
    blobs = np.full((F, C, m_max, 2), np.nan, dtype=np.float32)
    counts = np.zeros((F, C), dtype=np.int32)
    ident = np.full((F, C, m_max), -1, dtype=np.int32)
    w, h = rig["image_size"]
    for c in range(C):
        Xc = X0 @ rig["R"][c].T + rig["t"][c]
        K = rig["K"][c]
        u = K[0, 0] * Xc[..., 0] / Xc[..., 2] + K[0, 2] + rng.normal(0, noise_px, (F, M))
        v = K[1, 1] * Xc[..., 1] / Xc[..., 2] + K[1, 2] + rng.normal(0, noise_px, (F, M))
        if truncate:
            u, v = np.trunc(u), np.trunc(v)
        seen = (rng.random((F, M)) >= dropout) & (Xc[..., 2] > 0)
        seen &= (u >= 0) & (u < w) & (v >= 0) & (v < h)
        # random per-frame order, seen blobs first
        key = rng.random((F, M)) if shuffle else np.tile(np.arange(M, dtype=np.float64) / M, (F, 1))
        key = np.where(seen, key, 2.0)
        order = np.argsort(key, axis=1, kind="stable")
        n = seen.sum(axis=1).astype(np.int32)
        uu = np.take_along_axis(u, order, 1)
        vv = np.take_along_axis(v, order, 1)
        valid = np.arange(M)[None, :] < n[:, None]
        mm = min(M, m_max)
        blobs[:, c, :mm, 0] = np.where(valid, uu, np.nan)[:, :mm]
        blobs[:, c, :mm, 1] = np.where(valid, vv, np.nan)[:, :mm]
        ident[:, c, :mm] = np.where(valid, order, -1)[:, :mm]
        counts[:, c] = np.minimum(n, m_max)
    return blobs, counts, {"points_cam0": X0, "ident": ident}


def frame_to_reference_lists(blobs_f, counts_f, as_int=True):
    """One frame -> the nested lists `_find_dot` would produce (helpers.py:150-163):
    per camera a list of [x, y] Python ints, or [[None, None]] when the camera saw nothing."""
    out = []
    for c in range(blobs_f.shape[0]):
        n = int(counts_f[c])
        if n == 0:
            out.append([[None, None]])
            continue
        if as_int:
            out.append([[int(blobs_f[c, k, 0]), int(blobs_f[c, k, 1])] for k in range(n)])
        else:
            out.append([[float(blobs_f[c, k, 0]), float(blobs_f[c, k, 1])] for k in range(n)])
    return out


def make_ba_observations(rig, n_points, seed=0, noise_px=0.3, dropout=0.05, half_extent=0.7,
                         truncate=False):
    """Calibration capture set: obs f64 [N][C][2] with NaN for unseen (one marker waved around;
    computer_code/api/index.py:232 receives it as (N, C, 2) with None = unseen)."""
    rng = np.random.default_rng(seed)
    C = len(rig["R"])
    world = rng.uniform(-half_extent, half_extent, size=(n_points, 3))
    X0 = world @ rig["R0"].T + rig["centre"]
    obs = np.full((n_points, C, 2), np.nan)
    w, h = rig["image_size"]
    for c in range(C):
        Xc = X0 @ rig["R"][c].T + rig["t"][c]
        K = rig["K"][c]
        u = K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2] + rng.normal(0, noise_px, n_points)
        v = K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2] + rng.normal(0, noise_px, n_points)
        if truncate:
            u, v = np.trunc(u), np.trunc(v)
        seen = (rng.random(n_points) >= dropout) & (Xc[:, 2] > 0) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
        obs[seen, c, 0] = u[seen]
        obs[seen, c, 1] = v[seen]
    return obs, X0


def obs_to_reference_array(obs, as_int=False):
    """(N, C, 2) NaN-coded -> object ndarray with None, as np.array(cameraPoints) yields (index.py:232)."""
    N, C, _ = obs.shape
    out = np.empty((N, C, 2), dtype=object)
    for n in range(N):
        for c in range(C):
            if np.isnan(obs[n, c, 0]):
                out[n, c, 0] = None
                out[n, c, 1] = None
            else:
                out[n, c, 0] = int(obs[n, c, 0]) if as_int else float(obs[n, c, 0])
                out[n, c, 1] = int(obs[n, c, 1]) if as_int else float(obs[n, c, 1])
    return out


# Default to-world matrix of the reference UI (computer_code/src/App.tsx:45).  Numeric fixture only.
APP_TSX_TO_WORLD = [[0.9941338485260931, 0.0986512964608827, -0.04433748889242502, 0.9938296704767513],
                    [-0.0986512964608827, 0.659022672138982, -0.7456252673517598, 2.593331619023365],
                    [0.04433748889242498, -0.7456252673517594, -0.6648888236128887, 2.9576262456228286],
                    [0, 0, 0, 1]]


def make_object_frames(n_frames, k_max, seed=0, jitter=0.004):
    """World-coordinate point sets for the object locator (reference helpers.py:424-480): per frame
    0-2 drone LED triangles (0.095 / 0.095 / 0.15 m, jittered, random pose) among random clutter.
    Returns xyz f64 [F][k_max][3] (NaN padded), err f64 [F][k_max], n_pts i32 [F]."""
    rng = np.random.default_rng(seed)
    xyz = np.full((n_frames, k_max, 3), np.nan)
    err = np.full((n_frames, k_max), np.nan)
    n_pts = np.zeros(n_frames, dtype=np.int32)
    h = np.sqrt(0.095 ** 2 - 0.075 ** 2)
    for f in range(n_frames):
        pts = []
        for _ in range(int(rng.integers(0, 3))):
            c = rng.uniform(-1, 1, 3)
            ang = rng.uniform(0, 2 * np.pi)
            tilt = rng.normal(0, 0.2)
            u = np.array([np.cos(ang), np.sin(ang), tilt])
            u /= np.linalg.norm(u)
            w = np.cross(u, [0, 0, 1.0])
            w /= np.linalg.norm(w)
            side = 1.0 if rng.random() < 0.5 else -1.0
            tri = [c + side * h * w, c + 0.075 * u, c - 0.075 * u]
            pts += [p + rng.normal(0, jitter, 3) for p in tri]
        n_clutter = int(rng.integers(0, max(1, k_max - len(pts) + 1)))
        pts += [rng.uniform(-1, 1, 3) for _ in range(n_clutter)]
        pts = np.array(pts[:k_max]).reshape(-1, 3)
        order = rng.permutation(len(pts))
        pts = pts[order]
        n = len(pts)
        n_pts[f] = n
        xyz[f, :n] = pts
        err[f, :n] = rng.uniform(0.05, 2.0, n)
    return xyz, err, n_pts


# ----------------------------------------------------------------------------- camera frames (blob extraction)
REFERENCE_DISTORTION = (-1.26372388e-01, 2.62661497e-01, 1.21306197e-03, 2.24507008e-04, -2.48534118e-01)
"""distortion_coef of every entry of the reference's api/camera-params.json (k1 k2 p1 p2 k3)."""


def distort_pixels(u, v, K, dist):
    """Ideal pinhole pixel -> where the lens puts it (the forward model cv.undistort inverts)."""
    k1, k2, p1, p2, k3 = dist
    x = (u - K[0, 2]) / K[0, 0]
    y = (v - K[1, 2]) / K[1, 1]
    r2 = x * x + y * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]


def render_camera_frames(rig, n_frames, n_markers, seed=0, rows=240, cols=320, dist=REFERENCE_DISTORTION,
                         spot_sigma=(1.2, 2.2), noise_levels=3, dropout=0.05, half_extent=0.7, min_sep=0.12,
                         peak=400.0):
    """Synthetic PS3-Eye style IR frames: uint8 RGB [F][C][rows][cols][3], dark with sensor noise in
    [0, noise_levels) and one saturated Gaussian spot per visible marker at the lens-distorted
    projection.  K refers to the squared (cols x cols) frame the reference builds with make_square
    (helpers.py:72,507-523), so raw row = squared row - (cols - rows) // 2.
    Returns (frames, truth) with truth["uv"] [F][C][M][2] the ideal (undistorted) pixel, NaN = unseen."""
    rng = np.random.default_rng(seed)
    C = len(rig["R"])
    F, M = int(n_frames), int(n_markers)
    ay = (cols - rows) // 2
    world = _sample_markers(rng, F, M, half_extent, min_sep)
    X0 = world @ rig["R0"].T + rig["centre"]
    frames = rng.integers(0, max(1, noise_levels), (F, C, rows, cols, 3), dtype=np.uint8)
    uv = np.full((F, C, M, 2), np.nan)
    R = 9
    yy, xx = np.mgrid[-R:R + 1, -R:R + 1]
    for c in range(C):
        Xc = X0 @ rig["R"][c].T + rig["t"][c]
        K = rig["K"][c]
        u = K[0, 0] * Xc[..., 0] / Xc[..., 2] + K[0, 2]
        v = K[1, 1] * Xc[..., 1] / Xc[..., 2] + K[1, 2]
        ud, vd = distort_pixels(u, v, K, dist)
        vd = vd - ay
        seen = (rng.random((F, M)) >= dropout) & (Xc[..., 2] > 0)
        seen &= (ud >= 4) & (ud < cols - 4) & (vd >= 4) & (vd < rows - 4)
        sig = rng.uniform(spot_sigma[0], spot_sigma[1], (F, M))
        for f in range(F):
            img = frames[f, c].astype(np.float32)
            for m in np.flatnonzero(seen[f]):
                cx, cy = ud[f, m], vd[f, m]
                ix, iy = int(round(cx)), int(round(cy))
                x0, x1, y0, y1 = max(ix - R, 0), min(ix + R + 1, cols), max(iy - R, 0), min(iy + R + 1, rows)
                gx = xx[0, x0 - ix + R:x1 - ix + R] + ix - cx
                gy = yy[y0 - iy + R:y1 - iy + R, 0] + iy - cy
                spot = peak * np.exp(-(gy[:, None] ** 2 + gx[None, :] ** 2) / (2 * sig[f, m] ** 2))
                img[y0:y1, x0:x1] += spot[:, :, None]
                uv[f, c, m] = (u[f, m], v[f, m])
            frames[f, c] = np.clip(img, 0, 255).astype(np.uint8)
    return frames, {"uv": uv, "points_cam0": X0}
