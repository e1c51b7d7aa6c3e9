cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_cvo_data.py -m gpu -q -k "pipeline or rccl or pair_sharded or eval_cvo or cvo_scale or accflow_c1 or c3" > gpurun_out/r06/gputest_ks.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r06/gputest_ks.log | head -20
