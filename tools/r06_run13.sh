cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "encoder or norm or raft or stats" 2>&1 | grep -E "passed|failed"
tools/ab.sh "ACCFLOW_DIRECT_KT9=0" "" 2 --steps 16 2>&1 | tee gpurun_out/r06/ab_kt9_norm.txt
grep -E "^Cin(64|96|128) Cout(64|96|128) k3x3 s1 B7" gpurun_out/ab/conv_shapes_A.txt gpurun_out/ab/conv_shapes_B.txt | tee -a gpurun_out/r06/ab_kt9_norm.txt
