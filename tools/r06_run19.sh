cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
ACCFLOW_DIRECT_KT1=0 bash tools/prof_1stream.sh r06kt1a --ofe gma --height 720 --width 1280 > /dev/null 2>&1
bash tools/prof_1stream.sh r06kt1b --ofe gma --height 720 --width 1280 > /dev/null 2>&1
grep -h "7, false, 0, 0\|7, false, 1, 1\|sum of all\|ksplit\|pack_rows" gpurun_out/r06kt1a_kernel_stats_bench_1stream.txt gpurun_out/r06kt1b_kernel_stats_bench_1stream.txt | cut -c1-160
