cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_stg2/libaccflow_hip.so" "" 2 --steps 16 2>&1 | tee gpurun_out/r06/ab_stagger2.txt
tools/ab.sh "ACCFLOW_HIP_LIB=tools/bin/lib_stg5/libaccflow_hip.so" "" 2 --steps 16 2>&1 | tee gpurun_out/r06/ab_stagger5.txt
grep -E "Cin256 Cout(256|128) k(1x5|5x1)" gpurun_out/ab/conv_shapes_A.txt gpurun_out/ab/conv_shapes_B.txt | tee -a gpurun_out/r06/ab_stagger5.txt
