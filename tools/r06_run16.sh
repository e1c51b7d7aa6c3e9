cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gputest_e.log 2>&1; grep -E "passed|failed" gpurun_out/r06/gputest_e.log | tail -2; grep -B30 "short test summary" gpurun_out/r06/gputest_e.log | head -60
tools/ab_r05.sh 2 --steps 20 2>&1 | tee gpurun_out/r06/ab_r05_g.txt
