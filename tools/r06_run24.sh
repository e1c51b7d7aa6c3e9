cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "encoder_schedules" > gpurun_out/r06/gputest_enc2.log 2>&1; tail -3 gpurun_out/r06/gputest_enc2.log
for i in 1 2 3; do
for n in 0 2 3; do
  ACCFLOW_ENCODER_STREAMS=$n python bench.py --no-strict --no-extra --no-cpu-baseline --no-parity --steps 16 > gpurun_out/r06/enc_$n.json 2> /dev/null
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads([l for l in open("gpurun_out/r06/enc_%s.json" % n).read().strip().splitlines() if l.startswith("{")][-1])
print("ACCFLOW_ENCODER_STREAMS=%s  %.3f ms/step  %.3f ms one-at-a-time" % (n, d["ms_per_step"], d["one_sequence_at_a_time"]["ms_per_step"]))
PY
done
done
