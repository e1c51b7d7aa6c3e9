# Run on the GPU box (gpurun): every profile / bench file of a round that DESIGN.md and profiles/README.md quote, into gpurun_out/ (copy the summaries to profiles/).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/collect_profiles.sh r06 > gpurun_out/r06/collect.log 2>&1
python bench.py > gpurun_out/r06_bench_local.json 2> gpurun_out/r06/bench_local.err
python bench.py --steps 20 --warmup 5 --no-extra --no-strict --no-cpu-baseline > gpurun_out/r06_bench_20steps.json 2> /dev/null
python bench.py --no-strict --no-extra --no-cpu-baseline --no-parity --steps 8 --dump-kernels gpurun_out/r06_conv_shapes.txt > /dev/null 2>&1
python tools/lookup_sweep.py --fused > gpurun_out/r06_lookup_sweep.txt 2>&1
python tools/lc1_bench.py 0 1 > gpurun_out/r06_lc1_bench.txt 2>&1
tail -c 1500 gpurun_out/r06_bench_local.json
