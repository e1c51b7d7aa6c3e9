#!/bin/bash
# Same-box A/B of the WHOLE round-6 build against the round-5 tree (tools/bin/r05_tree: `git archive` of the round-5 HEAD with
# its own libaccflow_hip.so; built artefacts travel with gpurun).   tools/ab_r05.sh [rounds] [extra bench args]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R="${1:-2}"; shift || true
for i in $(seq 1 $R); do
  for v in r05 r06; do
    if [ $v = r05 ]; then D=tools/bin/r05_tree; else D=.; fi
    ( cd $D && timeout 600 python bench.py --no-strict --no-extra --no-cpu-baseline --steps 10 "$@" 2> /dev/null ) | python -c "
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$v  %.3f ms/step  %.1f TFLOP/s conv  %.3f ms one-at-a-time  EPE %.2e' % (d['ms_per_step'], d['roofline']['achieved'], d['one_sequence_at_a_time']['ms_per_step'], (d.get('parity') or {}).get('epe_mean_px', float('nan'))))
"
  done
done
