#!/bin/bash
# VERDICT r05 #7 - the one untried cell of the precision matrix: ONLY the direct kernel's S16 units (the update block: motion
# encoder, GRU, flow / mask heads; plus the few chain convolutions that run on them) on 2 of the fp16 split's 3 products;
# encoders, fusion chain (multi-source kernel), the fused lookup -> convc1 kernel and the GMA aggregation untouched.
#   tools/precision_probe_update.sh build   -> tools/bin/lib_upm{6,5}/libaccflow_hip.so   (CPU container)
#   tools/precision_probe_update.sh run     -> C3 (and with "run c5" also C5) EPE vs the reference's outputs + ms/step (GPU box)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
if [ "${1:-build}" = build ]; then
  python -m accflow_amd.build > /dev/null
  for m in 6 5; do
    d=tools/bin/lib_upm$m
    mkdir -p $d
    rm -rf $d/obj && cp -a accflow_amd/lib/obj $d/obj
    python -m accflow_amd.build --libdir=$d --unit-define=conv2d_direct_v_s16:ACCFLOW_F16_PAIRMASK=$m | tail -1
  done
else
  for m in 7 6 5; do
    lib=accflow_amd/lib/libaccflow_hip.so; [ $m != 7 ] && lib=tools/bin/lib_upm$m/libaccflow_hip.so
    ACCFLOW_HIP_LIB=$lib timeout 900 python bench.py --no-strict --no-extra --no-cpu-baseline --steps 8 2> /dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
p = d['parity']
print('C3 update-block PAIRMASK=$m (bit0 w_lo*x_hi, bit1 w_hi*x_lo, bit2 w_hi*x_hi): EPE vs reference mean %.2e max %.2e px | %.3f ms/step | one-at-a-time %.3f ms' % (p['epe_mean_px'], p['epe_max_px'], d['ms_per_step'], d['one_sequence_at_a_time']['ms_per_step']))
"
    if [ "${2:-}" = c5 ]; then
      ACCFLOW_HIP_LIB=$lib timeout 900 python bench.py --ofe gma --height 720 --width 1280 --no-strict --no-extra --no-cpu-baseline --steps 3 --warmup 1 2> /dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
p = d['parity']
print('C5 update-block PAIRMASK=$m: EPE vs reference mean %.2e max %.2e px | %.3f ms/step' % (p['epe_mean_px'], p['epe_max_px'], d['ms_per_step']))
"
    fi
  done
fi
