cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_backward.py tests/test_train.py -m gpu -x -q > gpurun_out/r06/gputest_wg.log 2>&1; tail -3 gpurun_out/r06/gputest_wg.log
for i in 1 2; do
  ACCFLOW_WGRAD_R64=0 python tools/wgrad_shapes.py --out gpurun_out/r06/wgrad_shapes_r64off.txt 2>&1 | tail -1
  python tools/wgrad_shapes.py --out gpurun_out/r06/wgrad_shapes_r64on.txt 2>&1 | tail -1
done
for i in 1 2; do
  ACCFLOW_WGRAD_R64=0 python tools/train_bench.py --steps 5 --warmup 2 2>&1 | tail -2 | cut -c1-300
  python tools/train_bench.py --steps 5 --warmup 2 2>&1 | tail -2 | cut -c1-300
done
