"""Self-launch of the one-process-per-GPU benchmarks.

`python bench.py --gpus N` as typed (no torch.distributed.run in front) must work: when N > 1 and the
process was not started by a launcher (WORLD_SIZE unset), start the N ranks as CHILD processes through
`python -m torch.distributed.run` on 127.0.0.1 and pass their output and exit code on.  This happens
before anything touches the GPU (no torch import here): a process that has initialised HIP must never
be replaced by another program.
"""
import os
import subprocess
import sys


def launched_by_torchrun():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def self_launch(script, argv, gpus):
    """Returns None when this process is a rank itself (N == 1, or started by a launcher); else runs the
    ranks as children and returns their exit code."""
    if gpus <= 1 or launched_by_torchrun():
        return None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    # --standalone: torchrun's own c10d rendezvous on a port IT binds (a port found here with bind(0) and closed again
    # could be taken by another process before torchrun binds it); --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={gpus}", os.path.abspath(script)] + list(argv)
    return subprocess.call(cmd, env=env)
