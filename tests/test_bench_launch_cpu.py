"""bench.py --gpus N must start its own ranks (the driver's command line is the only contract the multi-GPU path has:
the reference has no multi-anything).  No GPU here: MOCAP_BENCH_DRY=1 keeps the launcher, the rank
bookkeeping and the compact count-first exchange (gloo, two transfers in flight) and swaps the kernels for made-up
track records."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=240)


def test_bench_gpus2_self_launches_and_prints_one_rank0_line():
    p = _run({"MOCAP_BENCH_DRY": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "24")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["value"] is None
    ex = line["config"]["exchange"]
    assert ex["payload_checksums_match"] is True and ex["records_sent_all_ranks"] > 0
    assert line["config"]["frames_per_rank"] == [24, 25]  # uneven shards went through the count-first exchange


def test_bench_gpus2_dry_run_at_the_stress_shape():
    """--workload 64x256 through the same launcher / exchange: 64-camera records (160 bytes each)."""
    p = _run({"MOCAP_BENCH_DRY": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "9", "--workload", "64x256")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["config"]["workload"] == "64x256" and line["config"]["record_bytes"] == 160
    assert line["config"]["exchange"]["payload_checksums_match"] is True


def test_bench_world_size_mismatch_is_an_error_not_a_silent_single_rank_run():
    p = _run({"MOCAP_BENCH_DRY": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_profile_figures_are_marked_stale_when_the_sources_moved(monkeypatch):
    """roofline.traffic and roofline_fp64 are scaled from counter summaries under profiles/ (counters cannot be read inside
    the timed run).  Each summary names the sources it was taken on; on any other sources bench.py must say `stale`."""
    sys.path.insert(0, ROOT)
    import bench
    mix, stale, path = bench.load_profile("fp64_mix")
    assert mix is not None and path.startswith("profiles/") and "kernel_source_sha16" in mix and "kernel" in mix
    traffic, meta = bench.measured_traffic(1000)
    assert traffic > 0 and meta["traffic_stale"] is bool(stale) and meta["traffic_measured_on"]["kernel_source_sha16"]
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "0" * 16)
    assert bench.load_profile("fp64_mix")[1] is True
    assert bench.measured_traffic(1000)[1]["traffic_stale"] is True
    fp = bench.executed_fp64(1_000_000, 1.0)
    assert fp is None or fp.get("stale") is True
