#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run2
mkdir -p $O
timeout 900 python -m pytest tests/test_s16.py -x -q -s > $O/pytest_s16.log 2>&1; echo "s16 rc $?"; tail -8 $O/pytest_s16.log
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-extra --no-strict > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
ACCFLOW_S16=0 timeout 600 python bench.py --no-cpu-baseline --no-extra --no-strict > $O/bench_s16off.json 2> $O/bench_s16off.err; echo "bench rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_run2/bench*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["value"], d.get("parity"), d.get("roofline",{}).get("achieved"), d.get("roofline",{}).get("ms_per_step_in_kernel"), d.get("roofline_lookup",{}).get("frac"), d.get("one_sequence_at_a_time",{}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
