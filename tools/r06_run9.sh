cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_ah1/libaccflow_hip.so" "" 3 --steps 16 2>&1 | tee gpurun_out/r06/ab_gru_ahead.txt
grep -E "Cin256 Cout(256|128) k(1x5|5x1)" gpurun_out/ab/conv_shapes_A.txt | tee -a gpurun_out/r06/ab_gru_ahead.txt
grep -E "Cin256 Cout(256|128) k(1x5|5x1)" gpurun_out/ab/conv_shapes_B.txt | tee -a gpurun_out/r06/ab_gru_ahead.txt
tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_ah3/libaccflow_hip.so" "" 2 --steps 16 2>&1 | tee gpurun_out/r06/ab_gru_ahead3.txt
