"""A rank that never finishes (tests/test_launch.py: the launcher must not leave it behind)."""
import os
import sys
import time

print("PID %d rank %s cpus %s" % (os.getpid(), os.environ.get("RANK"), sorted(os.sched_getaffinity(0))), flush=True)
sys.stderr.flush()
time.sleep(10 ** 6)
