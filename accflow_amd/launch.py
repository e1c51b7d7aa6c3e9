"""One process per GPU without torchrun: spawn N ranks of a worker script and relay rank 0's output.

`bench.py --gpus N` (no WORLD_SIZE in the environment) uses this to start its own ranks, replacing
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


def rank_cpus(rank, world):
    """CPU set of one rank: the cores this process may use, dealt in contiguous blocks (at least one core per rank).  The
    child pins itself with os.sched_setaffinity before it imports torch (no GPU call has happened yet), so that the ranks'
    launch threads - 8.5 ms of Python per 26 ms step each - do not migrate over one another's cores."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:      # not Linux
        return None
    if not cpus or world <= 1:
        return None
    per = max(1, len(cpus) // world)
    lo = (rank * per) % len(cpus)
    return cpus[lo:lo + per]


def _pin(cpus):
    if cpus:
        os.sched_setaffinity(0, cpus)


def _stop(procs, grace_s=10.0):
    """Terminate, then kill, exactly the processes started here that are still running."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            p.send_signal(signal.SIGTERM)
        except OSError:
            pass
    deadline = time.time() + grace_s
    for p in live:
        try:
            p.wait(max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


DEFAULT_TIMEOUT_S = 3600.0


def spawn_ranks(argv, world, python=None, timeout=DEFAULT_TIMEOUT_S, poll_s=0.05, pin=True):
    """Run `python argv...` as `world` rank processes; rank 0 inherits stdout (its JSON line is the job's output), the
    other ranks' stdout goes to stderr.  Returns 0 if every rank exited 0; otherwise the first non-zero exit code (124
    after `timeout` seconds - finite by default, so that a collective no rank leaves cannot hang the job for ever), after
    the remaining ranks (exactly the PIDs started here) were terminated.  The ranks never outlive this call: whatever ends
    it - a failing rank, the timeout, an exception while starting a later rank, SIGTERM / SIGINT delivered to this
    process (`timeout 300 python bench.py --gpus N`) - every started rank is terminated, then killed."""
    python = python or sys.executable
    port = free_port()
    procs = []
    caught = []

    def on_signal(signum, frame):
        caught.append(signum)

    saved = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            saved[sg] = signal.signal(sg, on_signal)
        except ValueError:          # not the main thread: the try / finally below still cleans up
            pass
    rc = 0
    try:
        for r in range(world):
            out = None if r == 0 else sys.stderr
            cpus = rank_cpus(r, world) if pin else None
            procs.append(subprocess.Popen([python] + list(argv), env=rank_env(r, world, port), stdout=out,
                                          preexec_fn=(lambda c=cpus: _pin(c)) if cpus else None))
        t0 = time.time()
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
            if caught and rc == 0:
                rc = 128 + caught[0]
                print("launch: signal %d; stopping the ranks" % caught[0], file=sys.stderr)
            if rc == 0 and timeout is not None and time.time() - t0 > timeout:
                rc = 124
                print("launch: timeout after %.0f s" % timeout, file=sys.stderr)
            if rc != 0:
                break
            if live:
                time.sleep(poll_s)
    finally:
        _stop(procs)
        for sg, h in saved.items():
            signal.signal(sg, h)
    return rc
