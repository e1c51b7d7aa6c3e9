#!/bin/bash
# kernel-trace stats of the bench, one stream, one sequence at a time (per-launch averages readable from the table)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04a}
shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export ACCFLOW_STREAMS=1
B="python3 bench.py --no-cpu-baseline --no-parity --no-strict --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- $B --no-pipeline --steps 3 --warmup 1 > $OUT/bench_under_rocprof_1stream.log 2>&1
python3 - "$OUT" "$TAG" "$*" <<'PY'
import sys, os, glob, csv
out, tag, extra = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(os.path.join(out, "trace1", "*", "*kernel_stats.csv"))
rows = list(csv.DictReader(open(f[0])))
lines = ["ACCFLOW_STREAMS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --no-pipeline --steps 3 --warmup 1 --no-strict --no-extra %s (ONE stream, one sequence at a time; 6 sequence evaluations)" % extra,
         "%-100s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%")]
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"]) / 1e3
for r in rows[:45]:
    lines.append("%-100s %7s %12.1f %12.2f %7.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
lines.append("sum of all kernels: %.1f us = %.3f ms per sequence evaluation (6)" % (tot, tot / 6e3))
open("gpurun_out/%s_kernel_stats_bench_1stream.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
