"""GPU: the N-rank path end to end with real GPU shards.  The builder's boxes have one GPU and RCCL refuses two ranks on a
device, so the ranks share cuda:0 and talk over gloo -- what is exercised is everything but the transport: launcher,
contiguous uneven shards, the frame kernel per shard, on-device compaction, the count-first point-to-point exchange with
transfers in flight and buffer reuse, and BITWISE equality of the gathered tracks with the one-rank result."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,comm_stream,workload", [(2, False, "8x16"), (3, False, "8x16"), (2, True, "8x16"), (2, True, "64x256")])
def test_sharded_gpu_run_equals_one_rank_bitwise(world, comm_stream, workload):
    """comm_stream: the exchange is posted under a side stream and completed on a busy compute stream (bench.py's
    arrangement; the root's receive buffers then belong to the side stream's allocator pool).  64x256: BASELINE
    configs[4]'s shape (wide variant, 160-byte records)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MULTIRANK_COMM_STREAM"] = "1" if comm_stream else "0"
    env["MULTIRANK_WORKLOAD"] = workload
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_multirank_worker.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert f"MULTIRANK OK world {world}" in p.stdout


def test_bench_gpus2_starts_its_own_ranks_on_the_gpu():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it): one rank-0 line, n_gpus 2, the
    exchange really carried records.  (gloo + a shared device: a functional run, never a measurement.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MOCAP_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "2000", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-ba", "--no-blobs", "--no-latency"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["overflow_frames"] == 0
    ex = line["config"]["exchange"]
    assert ex["bytes_per_rank_per_step"] > 2000 * 20 * 48            # ~23 points x 48 bytes per frame
    assert 20 < line["config"]["markers_per_frame"] < 26
    # round 6: who took part (two ranks, ONE device here: the identities gathered over the process group say so -- a driver-side
    # scaling run proves N distinct GPUs by N distinct entries), and every rank's whole timed shard against the exhaustive walk
    assert line["world_size"] == 2 and line["backend"] == "gloo" and len(line["devices"]) == 2 and line["distinct_devices"] == 1
    assert line["parity"]["frames_checked"] == 4000 and line["parity"]["full_batch_vs_exhaustive_bit_exact"] is True
