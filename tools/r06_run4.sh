cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16.py tests/test_flowhead_fused.py -x -q -m gpu 2>&1 | tail -3
ACCFLOW_DIRECT_KT9=1 timeout 900 python -m pytest tests/test_s16.py tests/test_flowhead_fused.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "update_block or gru or raft or c2 or c3" 2>&1 | tail -3
tools/ab.sh "ACCFLOW_DIRECT_KT9=1" "" 3 --steps 16 2>&1 | tee gpurun_out/r06/ab_kt9.txt
tools/ab_r05.sh 2 2>&1 | tee gpurun_out/r06/ab_r05_c.txt
