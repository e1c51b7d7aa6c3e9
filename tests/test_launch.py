"""bench.py --gpus N without torchrun: the launcher (accflow_amd/launch.py) starts N rank processes BEFORE any GPU
call, relays rank 0's JSON line and returns the ranks' exit status.  CPU: a stub worker on gloo at N = 2 and the
error paths; GPU: bench.py itself through the launcher in the strong-scaling (pair-sharded) mode."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "workers", "stub_rank.py")


def _run_launcher(extra, world=2):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from accflow_amd.launch import spawn_ranks\n"
            "assert 'torch' not in sys.modules          # the parent stays GPU-free: it does not even import torch\n"
            "sys.exit(spawn_ranks([%r] + %r, %d, timeout=150))\n" % (ROOT, STUB, list(extra), world))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200)


@pytest.mark.timeout(240)
def test_launcher_spawns_ranks_and_relays_rank0():
    r = _run_launcher(["--tag", "x"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "rank 1 stdout" not in r.stdout     # only rank 0's stdout is the job's stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["master"] == "127.0.0.1" and d["args"] == ["--tag", "x"]
    assert "rank 1 stdout" in r.stderr


@pytest.mark.timeout(240)
def test_launcher_propagates_a_failing_rank():
    r = _run_launcher(["--fail-rank", "1"])
    assert r.returncode == 7 and "rank 1 exited with 7" in r.stderr


@pytest.mark.timeout(120)
def test_launcher_never_leaves_ranks_behind():
    """a parent that is killed (`timeout 300 python bench.py --gpus N`), a timeout, or a hung collective must
    not leave rank processes holding their GPUs.  The ranks here sleep for ever and print their PID; the parent gets
    SIGTERM (first case) or runs into its own timeout (second case); afterwards none of the PIDs exists."""
    import signal
    import time
    sleeper = os.path.join(ROOT, "tests", "workers", "sleep_rank.py")
    for mode in ("sigterm", "timeout"):
        code = ("import sys; sys.path.insert(0, %r)\n"
                "from accflow_amd.launch import spawn_ranks\n"
                "sys.exit(spawn_ranks([%r], 2, timeout=%s))\n" % (ROOT, sleeper, "None" if mode == "sigterm" else "3"))
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            so, se = os.path.join(td, "out"), os.path.join(td, "err")
            with open(so, "w") as fo, open(se, "w") as fe:
                p = subprocess.Popen([sys.executable, "-c", code], stdout=fo, stderr=fe)
            pids = []
            t0 = time.time()
            while len(pids) < 2 and time.time() - t0 < 60:      # rank 0 prints to stdout, rank 1 to the parent's stderr
                pids = [int(l.split()[1]) for f in (so, se) for l in open(f).read().splitlines() if l.startswith("PID ")]
                time.sleep(0.1)
            assert len(pids) == 2, pids
            if mode == "sigterm":
                p.send_signal(signal.SIGTERM)
            rc = p.wait(60)
            assert rc == (128 + signal.SIGTERM if mode == "sigterm" else 124), rc
            for pid in pids:
                with pytest.raises(ProcessLookupError):
                    os.kill(pid, 0)


def test_rank_cpu_sets_partition_the_allowed_cores():
    from accflow_amd.launch import rank_cpus
    allowed = sorted(os.sched_getaffinity(0))
    for world in (2, 4, 8):
        sets = [rank_cpus(r, world) for r in range(world)]
        assert all(s and set(s) <= set(allowed) for s in sets)
        if len(allowed) >= world:                      # disjoint blocks when there is at least one core per rank
            assert len(set().union(*map(set, sets))) == sum(map(len, sets))
    assert rank_cpus(0, 1) is None


def test_bench_self_launch_is_decided_before_any_gpu_call():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns; each child fails loudly on this GPU-less
    container (no CPU path), and the parent reports that instead of timing one rank as round 2 did."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.device_count() >= 2:
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    else:
        assert r.returncode != 0 and "needs GPU" in r.stderr and "launch: rank" in r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_rejects_gpus_world_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE 2" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_pairs_mode_through_launcher():
    """bench.py --gpus 1 --spawn --shard pairs: the launcher path and the strong-scaling mode (forward_pair_sharded over
    an RCCL group) end to end on the one GPU a box has; parity of its outputs vs the reference is in the JSON."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--shard", "pairs",
                        "--steps", "1", "--warmup", "1", "--no-strict"], capture_output=True, text=True, timeout=800, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["scaling"] == "strong"
    assert "pair-sharded" in d["config"]["parallelism"] and d["value"] > 0
    assert d["parity"]["epe_mean_px"] <= 1e-3
    print("pairs mode, 1 rank: %.2f ms/step" % d["ms_per_step"])
