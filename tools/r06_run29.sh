cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
ACCFLOW_PIPELINE_CTX_EARLY=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "sequence_pipeline" > gpurun_out/r06/gputest_ctx.log 2>&1; grep -E "passed|failed" gpurun_out/r06/gputest_ctx.log
bash tools/ab.sh "" "ACCFLOW_PIPELINE_CTX_EARLY=1" 3 --steps 16 > gpurun_out/r06/ab_ctx_early.txt 2>&1; cat gpurun_out/r06/ab_ctx_early.txt
