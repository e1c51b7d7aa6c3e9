cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16m.py tests/test_hip_parity.py -x -q -m gpu -k "statistics or instance or encoders or fnet or projection" 2>&1 | grep -E "passed|failed"
ACCFLOW_HIP_LIB=tools/bin/lib_bup/libaccflow_hip.so timeout 900 python -m pytest tests/test_s16.py tests/test_s16m.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_bup/libaccflow_hip.so" "" 3 --steps 16 2>&1 | tee gpurun_out/r06/ab_bupfront.txt
