"""One process per GPU without torchrun: spawn N ranks of a worker script and relay rank 0's output.

`bench.py --gpus N` (no WORLD_SIZE in the environment) and `eval_cvo` use this to start their own ranks, replacing
the reference's single-process nn.DataParallel (test_cvo.py:18,26).  The parent must not have touched the GPU: it
only parses arguments, starts the children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as
torch.distributed.run would set them) and waits - it never re-executes itself.  This module imports neither torch
nor the kernel library.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    # the host threads of N ranks share the box's cores: keep every rank's CPU pools small (the ranks only enqueue)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, world))))
    return env


def spawn_ranks(argv, world, python=None, timeout=None, poll_s=0.05):
    """Run `python argv...` as `world` rank processes; rank 0 inherits stdout (its JSON line is the job's output), the
    other ranks' stdout goes to stderr.  Returns 0 if every rank exited 0; otherwise the first non-zero exit code, after
    the remaining ranks (exactly the PIDs started here) were terminated."""
    python = python or sys.executable
    port = free_port()
    procs = []
    for r in range(world):
        out = None if r == 0 else sys.stderr
        procs.append(subprocess.Popen([python] + list(argv), env=rank_env(r, world, port), stdout=out))
    t0 = time.time()
    rc = 0
    live = set(range(world))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                print("launch: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
        if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
            if rc == 0:
                rc = 124
                print("launch: timeout after %.0f s" % timeout, file=sys.stderr)
            for r in sorted(live):
                procs[r].send_signal(signal.SIGTERM)
            deadline = time.time() + 10
            for r in sorted(live):
                try:
                    procs[r].wait(max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        if live:
            time.sleep(poll_s)
    return rc
