cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_s16.py tests/test_hip_parity.py -m gpu -x -q -k "gma or aggregat or chain or accflow_c1 or accplus or c5 or c3" > gpurun_out/r06/gputest_kt1.log 2>&1; tail -3 gpurun_out/r06/gputest_kt1.log
{ ACCFLOW_DIRECT_KT1=0 python tools/aggregate_bench.py --s16-only; python tools/aggregate_bench.py --s16-only; ACCFLOW_DIRECT_KT1=0 python tools/aggregate_bench.py --s16-only; python tools/aggregate_bench.py --s16-only; } 2>&1 | grep "S16 aggregation" > gpurun_out/r06/agg_bench_kt1.txt; cat gpurun_out/r06/agg_bench_kt1.txt
bash tools/ab.sh "ACCFLOW_DIRECT_KT1=0" "" 2 --ofe gma --height 720 --width 1280 --steps 3 --warmup 1 > gpurun_out/r06/ab_kt1_c5.txt 2>&1; cat gpurun_out/r06/ab_kt1_c5.txt
