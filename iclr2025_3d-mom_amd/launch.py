"""One process per GPU without torchrun: `bench.py --gpus N` (and any other entry point) starts its own ranks.

The parent NEVER touches the GPU (no HIP call, no torch.cuda.is_available()): on this pool a process that has initialised
the GPU must not be replaced, and a parent that held a context would also hold memory the ranks need.  It only forks N
children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (the variables torch.distributed's env://
rendezvous reads, the same ones `python -m torch.distributed.run` sets), relays rank 0's stdout and fails if any child fails.
"""
import os
import socket
import subprocess
import sys
import tempfile
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_by_a_launcher():
    """True inside a rank started by torch.distributed.run or by spawn_ranks()."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def spawn_ranks(n, argv, extra_env=None, timeout=None, poll_s=0.05):
    """Run `argv` n times, rank r with RANK=r, LOCAL_RANK=r, WORLD_SIZE=n.  Returns (exit code, rank 0's stdout).
    The exit code is 0 only if every rank exited 0; the first failing rank's code otherwise, after the remaining ranks
    have been terminated (a rank that died would leave the others waiting in a collective for ever).  stderr of every rank
    goes to this process's stderr."""
    port = free_port()
    procs = []
    out_file = tempfile.TemporaryFile(mode="w+")           # rank 0's stdout (a file cannot fill up like a pipe)
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=out_file if r == 0 else subprocess.DEVNULL, stderr=None))
    t0, rc = time.time(), 0
    live = set(range(n))
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
                if rc == 0:
                    rc = 124
                break
            if live:
                time.sleep(poll_s)
    finally:
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        out_file.seek(0)
        out0 = out_file.read()
        out_file.close()
    return rc, out0


def main_or_spawn(n, script, args, force=False):
    """Call at the top of an entry point, BEFORE any GPU call.  If this process is already a rank (started by a launcher) or
    a single process is wanted, returns False and the caller carries on.  Otherwise starts n ranks of the same command,
    prints rank 0's stdout, and exits with their status."""
    if launched_by_a_launcher() or (n <= 1 and not force):
        return False
    # MOM_SPAWN_TIMEOUT_S: the whole job's wall-clock limit (unset: none) -- past it the ranks are terminated, then killed, and the
    # parent exits 124; a rank that fails ends the others at once either way (spawn_ranks)
    limit = os.environ.get("MOM_SPAWN_TIMEOUT_S")
    rc, out = spawn_ranks(n, [sys.executable, script] + list(args), timeout=float(limit) if limit else None)
    sys.stdout.write(out or "")
    sys.stdout.flush()
    sys.exit(rc)
