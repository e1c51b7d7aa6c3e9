set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16m.py -x -q -m gpu 2>&1 | tail -5
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
tools/ab.sh "ACCFLOW_FUSE_PROJECTION=0" "" 2 2>&1 | tee gpurun_out/r06/ab_fuse_projection.txt
tools/precision_probe_update.sh run c5 2>&1 | tee gpurun_out/r06/precision_probe_update.txt
timeout 300 python tools/lookup_sweep.py --fused 2>&1 | tee gpurun_out/r06/lookup_sweep.txt
