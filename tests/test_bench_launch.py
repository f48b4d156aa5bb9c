"""bench.py starts its own ranks: `python bench.py --gpus N` with no launcher around it re-executes under
torch.distributed.run (one process per GPU) and never prints a line whose n_gpus differs from --gpus."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_by_itself():
    r = _run(["--gpus", "2", "--dry"], {"CV_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    assert lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == 2 and lines[0]["counted_ranks"] == 2
    assert lines[0]["backend"] == "gloo"


def test_train_mode_spawns_too():
    r = _run(["--gpus", "2", "--dry", "--mode", "train"], {"CV_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][0]
    assert line["n_gpus"] == 2 and line["mode"] == "train"


def test_refuses_a_line_with_the_wrong_rank_count():
    r = _run(["--gpus", "2", "--dry"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in r.stderr
