cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_s16.py tests/test_hip_parity.py tests/test_host_logic.py -x -q -k "gma or aggregat or abi or export or c5" > gpurun_out/r06/gputest_pack.log 2>&1; grep -E "passed|failed" gpurun_out/r06/gputest_pack.log
bash tools/ab.sh "ACCFLOW_GMA_PACK_ONCE=0" "" 3 --ofe gma --height 720 --width 1280 --steps 4 --warmup 1 > gpurun_out/r06/ab_gma_pack_once.txt 2>&1; cat gpurun_out/r06/ab_gma_pack_once.txt
