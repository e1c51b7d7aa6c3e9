cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "sequence_pipeline or encoder_schedules" > gpurun_out/r06/gputest_prio.log 2>&1; grep -E "passed|failed" gpurun_out/r06/gputest_prio.log
bash tools/ab.sh "ACCFLOW_PIPELINE_GROUP_PRIORITY=0" "" 3 --steps 16 > gpurun_out/r06/ab_group_prio2.txt 2>&1; cat gpurun_out/r06/ab_group_prio2.txt
