cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16.py tests/test_s16m.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "update_block or gru or raft or instance" 2>&1 | tail -3
tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_kt4/libaccflow_hip.so" "" 3 2>&1 | tee gpurun_out/r06/ab_kt_waves4.txt
tools/ab_r05.sh 1 2>&1 | tee gpurun_out/r06/ab_r05_b.txt
