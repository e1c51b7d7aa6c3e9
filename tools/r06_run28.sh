cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/ab.sh "" "ACCFLOW_PIPELINE_SPLIT=1" 2 --ofe gma --height 720 --width 1280 --steps 4 --warmup 1 > gpurun_out/r06/ab_pipeline_split_c5.txt 2>&1; cat gpurun_out/r06/ab_pipeline_split_c5.txt
ACCFLOW_PIPELINE_SPLIT=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "sequence_pipeline or eval_cvo or c3 or pipeline" > gpurun_out/r06/gputest_split2.log 2>&1; tail -3 gpurun_out/r06/gputest_split2.log
