cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16.py -x -q -m gpu 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_flowhead_fused.py tests/test_lookup_fused.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
tools/ab.sh "ACCFLOW_H16_STATE=0" "" 3 --steps 16 2>&1 | tee gpurun_out/r06/ab_gru_packed.txt
grep -E "Cin256 Cout(256|128) k(1x5|5x1)" gpurun_out/ab/conv_shapes_A.txt gpurun_out/ab/conv_shapes_B.txt | tee -a gpurun_out/r06/ab_gru_packed.txt
