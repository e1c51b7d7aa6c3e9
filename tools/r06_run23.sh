cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "encoders_on_two_streams or sequence_pipeline or warm_start or accflow_c1 or c3 or pair_sharded or rccl" > gpurun_out/r06/gputest_enc.log 2>&1; tail -3 gpurun_out/r06/gputest_enc.log
bash tools/ab.sh "ACCFLOW_ENCODER_STREAMS=0" "" 3 --steps 16 > gpurun_out/r06/ab_encoder_streams.txt 2>&1; cat gpurun_out/r06/ab_encoder_streams.txt
