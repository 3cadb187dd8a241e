"""Starting one process per GPU without an external launcher.

The reference is a single process (PathTracer.c:486-672).  A multi-GPU render here is one process per GPU; they
are normally started by `python -m torch.distributed.run`, which sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT.  launch_ranks() does the same from a plain `python program.py --gpus N`: the parent, which must not
have touched the GPU, starts N children with those variables set, passes rank 0's standard output through and
fails if any rank fails.
"""
import os
import socket
import subprocess
import sys


def under_launcher():
    """True in a process that a launcher (torch.distributed.run or launch_ranks) started as one rank of several."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(argv, n_ranks, env=None, timeout=None):
    """Runs `argv` n_ranks times, rank r with RANK = LOCAL_RANK = r.  Returns (exit code, rank 0's stdout).  The
    other ranks' stdout is dropped, every rank's stderr goes to this process's stderr.  If a rank fails, the others
    are ended and its exit code (non-zero) is returned."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n_ranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=sys.stderr))
    rc = 0
    out = b""
    try:
        out, _ = procs[0].communicate(timeout=timeout)
        for p in procs:
            code = p.wait(timeout=timeout)
            if code != 0 and rc == 0:
                rc = code
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    return rc, out.decode(errors="replace")
