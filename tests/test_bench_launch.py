"""`python bench.py --gpus N` as typed must start its own ranks (the plain form exited non-zero
unless torch.distributed.run was put in front).  CPU only: --dry-run makes the ranks meet over gloo and check
the band bookkeeping; no pixel is computed (the product path has no CPU fallback)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, [json.loads(ln) for ln in lines]


def test_plain_invocation_starts_two_ranks():
    p, out = run_bench("--gpus", "2", "--dry-run", "--height", "1080")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(out) == 1                                         # ONE JSON line, from rank 0
    assert out[0]["n_gpus"] == 2 and out[0]["ranks_seen"] == 2 and out[0]["rows_covered_once"]
    assert out[0]["launched_by"] == "self"


def test_three_ranks_and_an_odd_height():
    p, out = run_bench("--gpus", "3", "--dry-run", "--height", "77")
    assert p.returncode == 0, p.stderr[-2000:]
    assert out[0]["ranks_seen"] == 3 and out[0]["rows_covered_once"]


def test_single_rank_does_not_launch():
    p, out = run_bench("--gpus", "1", "--dry-run")
    assert p.returncode == 0 and out[0]["n_gpus"] == 1 and out[0]["launched_by"] == "none"


def test_world_size_mismatch_is_an_error():
    p, _ = run_bench("--gpus", "2", "--dry-run", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
