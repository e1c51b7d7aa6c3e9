cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_lean/libaccflow_hip.so" "" 3 --steps 16 2>&1 | tee gpurun_out/r06/ab_direct_lean.txt
tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_noepi/libaccflow_hip.so" "" 1 --steps 16 --no-parity 2>&1 | tee gpurun_out/r06/abl_gru_noepi.txt
grep -E "k1x5|k5x1" gpurun_out/ab/conv_shapes_A.txt | tee -a gpurun_out/r06/abl_gru_noepi.txt
grep -E "k1x5|k5x1" gpurun_out/ab/conv_shapes_B.txt | tee -a gpurun_out/r06/abl_gru_noepi.txt
