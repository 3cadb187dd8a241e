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
import threading
import time


def under_launcher():
    """True in a process that a launcher (torch.distributed.run or launch_ranks) started as one rank of several."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(argv, n_ranks, env=None, timeout=None, poll_s=0.05):
    """Runs `argv` n_ranks times, rank r with RANK = LOCAL_RANK = r.  Returns (exit code, rank 0's stdout).  The
    other ranks' stdout is dropped, every rank's stderr goes to this process's stderr.

    All children are watched together: as soon as one of them exits with a non-zero code the others are ended and
    that code is returned (a rank that dies before the rendezvous would otherwise leave the rest waiting in
    init_process_group until the store's own timeout).  `timeout` bounds the whole launch in seconds (exit code 124
    when it runs out).  Children are always fresh processes; nothing is re-executed."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    chunks = []

    def drain(pipe):  # rank 0's stdout, read as it comes so that a full pipe never blocks the child
        for block in iter(lambda: pipe.read(65536), b""):
            chunks.append(block)

    reader = None
    rc = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        for r in range(n_ranks):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=sys.stderr))
        reader = threading.Thread(target=drain, args=(procs[0].stdout,), daemon=True)
        reader.start()
        running = set(range(n_ranks))
        while running and rc == 0:
            for r in sorted(running):
                code = procs[r].poll()
                if code is None:
                    continue
                running.discard(r)
                if code != 0:
                    rc = code
                    break
            if running and rc == 0:
                if deadline is not None and time.monotonic() > deadline:
                    rc = 124
                    break
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        if reader is not None:
            reader.join(timeout=10)
    return rc, b"".join(chunks).decode(errors="replace")
