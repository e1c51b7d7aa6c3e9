#!/bin/bash
# Same-box A/B of two builds / settings of the benchmark (boxes differ by +-3 %, so nothing is compared across gpurun calls):
#   tools/ab.sh "<env A>" "<env B>" [rounds] [extra bench args]     e.g.  tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_ref/libaccflow_hip.so" ""
# prints ms/step (pipelined), conv family TFLOP/s and ms/step one sequence at a time for every run.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
A="$1"; B="$2"; R="${3:-2}"; shift; shift; shift || true
mkdir -p gpurun_out/ab
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E timeout 600 python bench.py --no-strict --no-extra --no-cpu-baseline --no-parity --steps 8 "$@" --dump-kernels gpurun_out/ab/conv_shapes_$v.txt > gpurun_out/ab/bench_$v.json 2> gpurun_out/ab/bench_$v.err || echo "run $v failed"
    python - "$v" "$E" <<'PY'
import json, sys
v, e = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open("gpurun_out/ab/bench_%s.json" % v).read().strip().splitlines() if l.startswith("{")][-1])
    print("%s [%s]  %.3f ms/step  %.1f TFLOP/s conv  %.3f ms one-at-a-time" % (v, e, d["ms_per_step"], d["roofline"]["achieved"], d["one_sequence_at_a_time"]["ms_per_step"]))
except Exception as ex:
    print(v, "ERR", ex)
PY
  done
done
