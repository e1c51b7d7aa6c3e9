cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/ab.sh "" "ACCFLOW_GROUP_PRIORITY=-1" 3 --steps 16 > gpurun_out/r06/ab_group_prio.txt 2>&1; cat gpurun_out/r06/ab_group_prio.txt
