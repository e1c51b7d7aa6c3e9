cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gputest_b.log 2>&1; tail -4 gpurun_out/r06/gputest_b.log
tools/ab.sh "ACCFLOW_DIRECT_KT9=0" "" 2 --steps 16 2>&1 | tee gpurun_out/r06/ab_kt9_all.txt
tools/ab_r05.sh 2 2>&1 | tee gpurun_out/r06/ab_r05_d.txt
