cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16m.py -x -q -m gpu 2>&1 | tail -3
ACCFLOW_S16_VIA_MULTI=1 timeout 900 python -m pytest tests/test_s16.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
tools/ab.sh "ACCFLOW_S16M_KT9=0" "" 3 --steps 16 2>&1 | tee gpurun_out/r06/ab_s16m_kt9.txt
