#!/bin/bash
# Per-layer-group precision probe (VERDICT r04 #2d): experiment libraries in which ONLY the multi-source S16 kernel - every
# convolution of the three encoders and of the fusion chain - runs a subset of the fp16 split's three products, the update
# block (direct kernel) untouched.  Build here (CPU container), run on the GPU box:
#   tools/precision_probe_s16m.sh build        -> tools/bin/lib_pm{6,5,4}/libaccflow_hip.so
#   tools/precision_probe_s16m.sh run          -> EPE vs the reference's outputs + ms/step per library (on the GPU)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
if [ "${1:-build}" = build ]; then
  python -m accflow_amd.build > /dev/null
  for m in 6 5 4; do
    d=tools/bin/lib_pm$m
    mkdir -p $d
    rm -rf $d/obj && cp -a accflow_amd/lib/obj $d/obj
    python -m accflow_amd.build --libdir=$d --unit-define=conv_s16m_v:S16M_PAIRMASK=$m | tail -1
  done
else
  for m in 7 6 5 4; do
    lib=accflow_amd/lib/libaccflow_hip.so; [ $m != 7 ] && lib=tools/bin/lib_pm$m/libaccflow_hip.so
    ACCFLOW_HIP_LIB=$lib timeout 900 python bench.py --no-strict --no-extra --no-cpu-baseline --steps 8 2> /dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
p = d['parity']
print('S16M_PAIRMASK=$m (bit0 w_lo*x_hi, bit1 w_hi*x_lo, bit2 w_hi*x_hi): EPE vs reference mean %.2e max %.2e px | %.3f ms/step | one-at-a-time %.3f ms' % (p['epe_mean_px'], p['epe_max_px'], d['ms_per_step'], d['one_sequence_at_a_time']['ms_per_step']))
"
  done
fi
