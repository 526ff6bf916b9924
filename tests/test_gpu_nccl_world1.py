"""Multi-GPU readiness on a one-GPU box: the process-per-GPU paths run under `python -m torch.distributed.run
--nproc-per-node 1` with backend nccl (= RCCL), so that RCCL initialisation, dist.gather / dist.reduce / all_gather on
device tensors and the barriers execute on real hardware with world = 1 (plain `python bench.py` has no process group
and copies locally).  The ranks are CHILD processes; nothing here replaces a process that has touched the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def torchrun(script, *argv, timeout=240):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node=1", os.path.join(ROOT, script), *argv]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[-1])


def test_bench_tile_sharding_through_rccl_with_one_rank():
    d = torchrun("bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "640", "--height", "360", "--spp", "64",
                 "--no-extras")
    assert d["process_group"] == {"backend": "nccl", "world": 1}
    assert d["n_gpus"] == 1 and d["value"] > 0 and 1.0 < d["rays_per_sample"] < 3.0


def test_sample_range_reduce_through_rccl_with_one_rank():
    d = torchrun("tools/bench_spheres.py", "--gpus", "1", "--spheres", "20000", "--spp", "8", "--steps", "1", "--width", "320",
                 "--height", "180", "--cpu-spheres", "0")
    assert d["process_group"] == {"backend": "nccl", "world": 1}
    assert d["Msamples_per_s"] > 0 and max(d["mean_radiance"]) > 0


def test_photon_build_all_gather_through_rccl_with_one_rank():
    d = torchrun("tests/rank_photon_world1.py")
    assert d["backend"] == "nccl" and d["world"] == 1
    assert d["maps_equal"] and d["emitted_equal"] and d["reduce_is_identity"] and d["photons"] == 20000
