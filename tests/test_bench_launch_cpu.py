"""`python bench.py --gpus N` must start its own N ranks (the driver's scaling run is a plain
`python3 bench.py --gpus 8`).  --launch-check stops every rank before it touches a GPU, so this runs on CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_plain_invocation_launches_its_own_ranks():
    rec = _run(["--gpus", "2", "--launch-check", "1"])
    assert rec["n_gpus"] == 2 and rec["master"] == "127.0.0.1" and rec["backend"] == "nccl"


def test_under_a_launcher_it_is_one_rank():
    # WORLD_SIZE already set (torch.distributed.run did the launching): no second level of processes
    rec = _run(["--gpus", "3", "--launch-check", "1"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0",
                                                         "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert rec["n_gpus"] == 3
