cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for i in 1 2; do
for n in 0 1 2 3; do
  ACCFLOW_ENCODER_STREAMS=$n python bench.py --ofe gma --height 720 --width 1280 --no-pipeline --no-strict --no-extra --no-cpu-baseline --no-parity --steps 4 --warmup 1 > gpurun_out/r06/c5enc_$n.json 2> /dev/null
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads([l for l in open("gpurun_out/r06/c5enc_%s.json" % n).read().strip().splitlines() if l.startswith("{")][-1])
print("C5 one at a time ACCFLOW_ENCODER_STREAMS=%s  %.3f ms/step" % (n, d["ms_per_step"]))
PY
done
done
for p in 0 -1; do ACCFLOW_GROUP_PRIORITY=$p python bench.py --ofe gma --height 720 --width 1280 --no-pipeline --no-strict --no-extra --no-cpu-baseline --no-parity --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('C5 one at a time priority $p', d['ms_per_step'])"; done
