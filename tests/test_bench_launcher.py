"""bench.py's own launcher: `python bench.py --gpus N` with no WORLD_SIZE in the environment must start its N ranks
itself (the driver's command shape).  CPU box: the rank plumbing with gloo and a stubbed step
(PGORB_BENCH_LAUNCHER_TEST); GPU box: the real bench at N = 1, directly and through the self-launch path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    return p


def _last_json(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert lines, p.stderr[-3000:]
    return json.loads(lines[-1])                      # the JSON line is the LAST line of stdout


@pytest.mark.parametrize("world", [2, 4])
def test_bench_launches_its_own_ranks(world):
    p = _run(["--gpus", str(world), "--steps", "5", "--warmup", "0", "--batch", "8", "--n1-fps", "1000"],
             {"PGORB_BENCH_LAUNCHER_TEST": "1"}, 300)
    out = _last_json(p)
    assert out["launcher_test"] is True and out["n_gpus"] == world and out["steps"] == 5
    assert len(out["per_rank_fps"]) == world and out["rides"] == [[r] for r in range(world)]
    # rank r sleeps 2 (r + 1) ms per step: ranks get slower with r, and `value` follows the slowest one
    assert all(out["per_rank_fps"][r] > out["per_rank_fps"][r + 1] for r in range(world - 1))
    assert out["max_rank_seconds"] >= 5 * 0.002 * world
    assert abs(out["value"] - world * 8 * 5 / out["max_rank_seconds"]) < 1e-6 * out["value"]
    assert abs(out["scaling_efficiency"] - out["value"] / (world * 1000.0)) < 1e-9
    assert out["config"]["vocab_broadcast_bytes"] > 0


def test_launched_rank_rejects_a_mismatched_world():
    p = _run(["--gpus", "3", "--steps", "1"], {"PGORB_BENCH_LAUNCHER_TEST": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, 120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_bench_at_n1_directly_and_through_the_self_launch_path():
    quick = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-upload-leg",
             "--no-overlap-leg", "--sustain-seconds", "0"]
    a = _last_json(_run(quick, {}, 900))
    b = _last_json(_run(quick, {"PGORB_BENCH_FORCE_LAUNCH": "1", "PGORB_BENCH_FORCE_DIST": "1"}, 900))
    for out in (a, b):
        assert out["n_gpus"] == 1 and out["verified"] is True and out["value"] > 0
        assert len(out["per_rank_fps"]) == 1 and out["roofline"]["frac"] > 0
    assert b["config"]["vocab_broadcast_bytes"] > 0          # the RCCL branch ran in the launched rank
