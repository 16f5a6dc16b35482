"""TEST / BASELINE INFRASTRUCTURE (bench.py's cpu_baseline leg only): the NumPy/Python restatement of the reference's frame
path (oracle/mocap_oracle.py: the reference's own structure -- Python loops, one LAPACK SVD per candidate group -- bit-exact
against the reference-run goldens) timed on >= 200 frames spread over the host's cores, as BASELINE.md section 3 asks.
Workers are separate interpreters started with subprocess (`python -m oracle.port_pool <job.npz>`): nothing is forked from a
process that holds a HIP context, nothing re-imports the caller's main module; they import NumPy and the oracle only."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _work(path):
    import numpy as np
    from oracle import mocap_oracle as mo
    z = np.load(path)
    K, R, t, blobs, counts = z["K"], z["R"], z["t"], z["blobs"], z["counts"]
    Ks = [k for k in K]
    Ftab = mo.fundamental_table(Ks, R, t)
    c0, w0 = time.process_time(), time.perf_counter()
    pts = 0
    for f in range(blobs.shape[0]):
        o = mo.find_point_correspondance_and_object_points(blobs[f], counts[f], Ks, R, t, Ftab=Ftab)
        pts += len(o["errors"])
    return {"markers": pts, "frames": int(blobs.shape[0]), "cpu_s": time.process_time() - c0, "wall_s": time.perf_counter() - w0}


def python_port_rate(rig, blobs, counts, frames=200, max_workers=None, timeout_s=180):
    """-> dict: markers/s per core (markers / summed worker CPU seconds) and over all workers (markers / the slowest worker's
    wall time), frames, workers.  Frame i goes to worker i mod workers (candidate counts per frame are heavy-tailed)."""
    import numpy as np
    frames = int(min(frames, blobs.shape[0]))
    workers = int(max(1, min(max_workers or (os.cpu_count() or 1), frames, 64)))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    w0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for w in range(workers):
            idx = np.arange(w, frames, workers)
            path = os.path.join(d, f"job{w}.npz")
            np.savez(path, K=np.asarray(rig["K"]), R=np.asarray(rig["R"]), t=np.asarray(rig["t"]), blobs=blobs[idx], counts=counts[idx])
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.port_pool", path], cwd=ROOT, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        res = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=max(1.0, timeout_s - (time.perf_counter() - w0)))
                res.append(json.loads(out.strip().split("\n")[-1]))
            except Exception:
                p.kill()
    wall = time.perf_counter() - w0
    if not res:
        raise RuntimeError("no worker of the Python port finished")
    pts = sum(r["markers"] for r in res)
    return {"markers_per_s_per_core": pts / sum(r["cpu_s"] for r in res), "markers_per_s_all_workers": pts / max(r["wall_s"] for r in res),
            "frames": sum(r["frames"] for r in res), "workers": len(res), "wall_s_incl_startup": wall, "markers": pts}


if __name__ == "__main__":
    print(json.dumps(_work(sys.argv[1])))
