"""bench.py's launcher logic on the CPU (no HIP call, no stepping): `python bench.py --gpus 2` outside
torch.distributed.run must spawn the two ranks itself, shard ONE ensemble with shard_bounds under
--total-reactors, reduce the timing with MAX, gather unequal shards and print exactly one JSON line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    return p


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_self_launch_two_ranks_strong_scaling(wt):
    p = _run(["--gpus", "2", "--total-reactors", "20001", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout          # rank 0 only
    o = lines[0]
    assert o["n_gpus"] == 2 and o["scaling"] == "strong" and o["dry_run"] is True
    assert o["shard_sizes"] == [10001, 10000]   # shard_bounds: the first N % W ranks get one more
    assert o["config"]["reactors_total"] == 20001 and o["config"]["reactors_per_gpu"] == 10001
    # the gathered state is the whole ensemble in reactor order
    cols, _ = wt.make_ensemble(20001)
    expect = 8 * float(cols["initial_pH"].sum() + cols["initial_chlorine"].sum() + cols["temperature"].sum())
    assert abs(o["state_checksum"] - expect) <= 1e-9 * abs(expect)
    assert o["final_gather_ms"] is not None


def test_self_launch_weak_scaling_slices_are_distinct(wt):
    p = _run(["--gpus", "2", "--reactors", "300", "--zones", "4", "--steps", "2", "--warmup", "0", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    o = _json_lines(p.stdout)[0]
    assert o["scaling"] == "weak" and o["config"]["reactors_total"] == 600
    cols, _ = wt.make_ensemble(600)           # rank r owns reactors [300 r, 300 (r + 1)) of the population
    expect = 4 * float(cols["initial_pH"].sum() + cols["initial_chlorine"].sum() + cols["temperature"].sum())
    assert abs(o["state_checksum"] - expect) <= 1e-9 * abs(expect)


def test_single_rank_line_shape():
    p = _run(["--steps", "2", "--warmup", "0", "--reactors", "128", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    o = _json_lines(p.stdout)[0]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in o
    assert o["n_gpus"] == 1 and o["scaling"] == "weak" and "workload" in o["config"]


def test_failing_rank_fails_the_launcher():
    # 3 reactors over 2 ranks is fine, 1 reactor over 2 ranks leaves rank 1 empty -> that rank exits 2
    p = _run(["--gpus", "2", "--total-reactors", "1", "--steps", "1", "--warmup", "0", "--dry-run"])
    assert p.returncode != 0


def test_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "2", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2


def test_roofline_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    # 48 B state per zone-step; derived 24 B and boundary 80/n B move once per work item
    assert bench.algorithmic_bytes_per_zone_step(8, 1) == pytest.approx(48 + 24 + 10)
    assert bench.algorithmic_bytes_per_zone_step(8, 50) == pytest.approx(48 + 34 / 50)
    assert bench.algorithmic_bytes_per_zone_step(4, 5) == pytest.approx(48 + 44 / 5)


@pytest.mark.parametrize("sensors", [False, True], ids=["config4", "config5"])
def test_eight_rank_rehearsal_of_the_baseline_configs(wt, sensors):
    """BASELINE configs 4 and 5 exactly as the driver will launch them on a node -- 100 000 reactors x 8 zones cut over
    8 ranks (12 500 each), with and without the sensor suite -- rehearsed on the CPU over gloo: launcher, sharding,
    MAX-reduced timing, the padded all_gather and the single JSON line.  (No stepping: there is no GPU here.)"""
    args = ["--gpus", "8", "--total-reactors", "100000", "--steps", "2", "--warmup", "1", "--dry-run"] + (["--sensors"] if sensors else [])
    p = _run(args, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    o = lines[0]
    assert o["n_gpus"] == 8 and o["scaling"] == "strong" and o["dry_run"] is True
    assert o["shard_sizes"] == [12500] * 8
    assert o["config"]["reactors_total"] == 100000 and o["config"]["reactors_per_gpu"] == 12500 and o["config"]["zones"] == 8
    assert o["final_gather_ms"] is not None and o["final_gather_ms"] >= 0
    assert "8" in o["config"]["sharding"]
    cols, _ = wt.make_ensemble(100000)
    expect = 8 * float(cols["initial_pH"].sum() + cols["initial_chlorine"].sum() + cols["temperature"].sum())
    assert abs(o["state_checksum"] - expect) <= 1e-9 * abs(expect)     # the gathered block is the whole ensemble, in order
