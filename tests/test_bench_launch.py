"""bench.py's rank plumbing without a GPU: `python bench.py --gpus N` must start N ranks itself when no launcher
did, the ranks must find each other, and rank 0 alone prints the JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CAMSYNC_BENCH_SELFTEST"] = "1"
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]  # (gloo itself writes a "[Gloo] ..." line to stdout)
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["local_ranks"] == [0, 1] and out["max_dt"] == 2.0
    assert out["steps"] == 2 and out["warmup"] == 1


def test_launched_by_torchrun_style_env():
    """The driver's form: the launcher sets RANK/LOCAL_RANK/WORLD_SIZE; bench.py must not start ranks of its own."""
    r = _run(["--gpus", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29531"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_gpus_must_agree_with_world_size():
    r = _run(["--gpus", "4"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29532"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
