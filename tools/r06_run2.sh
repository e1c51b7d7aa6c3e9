set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16m.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -8
tools/ab_r05.sh 2 2>&1 | tee gpurun_out/r06/ab_r05_a.txt
