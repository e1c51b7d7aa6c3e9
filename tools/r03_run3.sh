#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run3
mkdir -p $O
timeout 900 python -m pytest tests/test_s16.py -x -q -s > $O/pytest_s16.log 2>&1; echo "s16 rc $?"; tail -8 $O/pytest_s16.log
timeout 1500 python -m pytest tests -m gpu -x -q -k "gma or c5 or C5" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
timeout 600 python bench.py --ofe gma --height 720 --width 1280 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-strict --no-extra > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_run3/bench*.json")):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["value"], d.get("parity"), d.get("roofline",{}).get("achieved"), d.get("one_sequence_at_a_time",{}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
