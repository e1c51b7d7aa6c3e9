cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "encoder_schedules or sequence_pipeline" > gpurun_out/r06/gputest_split.log 2>&1; tail -3 gpurun_out/r06/gputest_split.log
bash tools/ab.sh "" "ACCFLOW_PIPELINE_SPLIT=1" 3 --steps 16 > gpurun_out/r06/ab_pipeline_split.txt 2>&1; cat gpurun_out/r06/ab_pipeline_split.txt
