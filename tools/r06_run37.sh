cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
K="sequence_pipeline or encoder_schedules or accflow_c1 or c3_ or warm_start or pair_sharded or rccl or eval_cvo or mid_size"
for e in "ACCFLOW_STREAMS=1" "ACCFLOW_ENCODER_STREAMS=3 ACCFLOW_GROUP_PRIORITY=0" "ACCFLOW_PIPELINE_SPLIT=0 ACCFLOW_ENCODER_STREAMS=0" "ACCFLOW_H16_STATE=0 ACCFLOW_DIRECT_KT9=0 ACCFLOW_S16M_KT9=0" "ACCFLOW_FUSE_PROJECTION=0 ACCFLOW_DEFORM_S16_COLUMNS=0" "ACCFLOW_CONV_MODE=bf16x6"; do
  env $e timeout 900 python -m pytest tests/test_hip_parity.py tests/test_cvo_data.py -m gpu -x -q -k "$K" > gpurun_out/r06/gputest_env.log 2>&1
  echo "[$e] $(grep -E 'passed|failed|error' gpurun_out/r06/gputest_env.log | tail -1)"
done
