cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/ab_r05.sh 2 --steps 20 > gpurun_out/r06/ab_r05_final.txt 2>&1; cat gpurun_out/r06/ab_r05_final.txt
bash tools/ab.sh "ACCFLOW_ENCODER_STREAMS=0 ACCFLOW_PIPELINE_SPLIT=0 ACCFLOW_GROUP_PRIORITY=0" "" 2 --steps 20 > gpurun_out/r06/ab_schedules_total.txt 2>&1; cat gpurun_out/r06/ab_schedules_total.txt
