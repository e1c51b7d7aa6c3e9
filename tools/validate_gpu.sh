#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/validate
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/validate/bench.json").read().strip().splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["parity"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline_lookup"]["frac"], d["one_sequence_at_a_time"]["ms_per_step"])
print(d.get("other_configs"))
PY
python __graft_entry__.py smoke
