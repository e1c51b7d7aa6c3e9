#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run4
mkdir -p $O
python tools/lookup_sweep.py > $O/lookup_sweep.txt 2>&1; cat $O/lookup_sweep.txt
timeout 600 python bench.py --dump-kernels $O/conv_shapes.txt > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03_run4/bench.json").read().strip().splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["parity"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["frac_executed"], d["roofline_lookup"]["frac"])
print(d.get("other_configs"))
PY
head -30 $O/conv_shapes.txt
