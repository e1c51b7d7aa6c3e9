#!/bin/bash
# round 3, GPU call 1: the parity suite, the bench with the new launcher paths, and a precision probe of the fp16 split
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 300 python bench.py --gpus 1 --spawn --shard pairs --steps 5 --warmup 2 --no-strict > $O/bench_pairs.json 2> $O/bench_pairs.err; echo "pairs rc $?"
for m in 6 5 4; do
  ACCFLOW_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/lib_mask$m/libaccflow_hip.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strict --no-extra > $O/bench_mask$m.json 2> $O/bench_mask$m.err; echo "mask $m rc $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_run1/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d.get("parity"), d.get("roofline",{}).get("achieved"), d.get("roofline_lookup",{}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
