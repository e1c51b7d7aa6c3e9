#!/bin/bash
# rocprofv3 kernel-trace stats of the training step (tools/train_bench.py, eager steps only: --graph 0): 2 warm-up + 5 timed
# training steps + 6 inference forwards of the same batch.   tools/prof_train.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}
OUT=gpurun_out/prof_train_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/train_bench.py --graph 0 --steps 5 --warmup 2 > $OUT/train_under_rocprof.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import sys, os, glob, csv
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv"))
rows = list(csv.DictReader(open(f[0])))
lines = ["rocprofv3 --kernel-trace --stats -- python3 tools/train_bench.py --graph 0 --steps 5 --warmup 2  (7 eager training steps of AccFlow(RAFT) 7x256x256 batch 6 + 6 inference forwards of the same batch)",
         "%-100s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%")]
tot = sum(float(r["TotalDurationNs"]) / 1e3 for r in rows)
for r in rows[:40]:
    lines.append("%-100s %7s %12.1f %12.2f %7.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
lines.append("sum of all kernels: %.1f us" % tot)
lines.append(open(os.path.join(out, "train_under_rocprof.log")).read().strip().splitlines()[-1])
open("gpurun_out/%s_train_kernel_stats.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
