cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python bench.py --ofe gma --height 720 --width 1280 --steps 3 --warmup 1 --no-extra --no-strict --no-cpu-baseline --no-parity --dump-kernels gpurun_out/r06/c5_conv_shapes.txt > gpurun_out/r06/c5_pipe.json 2> gpurun_out/r06/c5_pipe.err
python bench.py --ofe gma --height 720 --width 1280 --steps 3 --warmup 1 --no-extra --no-strict --no-cpu-baseline --no-parity --no-pipeline > gpurun_out/r06/c5_one.json 2> gpurun_out/r06/c5_one.err
tail -c 600 gpurun_out/r06/c5_pipe.json; tail -c 600 gpurun_out/r06/c5_one.json
