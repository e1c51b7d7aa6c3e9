cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/ab.sh "" "ACCFLOW_CONV_KSPLIT=0" 3 --steps 16 > gpurun_out/r06/ab_ksplit_off.txt 2>&1; cat gpurun_out/r06/ab_ksplit_off.txt
