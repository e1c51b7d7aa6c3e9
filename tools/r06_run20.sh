cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest_f.log 2>&1; tail -2 gpurun_out/r06/gputest_f.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke_f.log 2>&1; tail -2 gpurun_out/r06/smoke_f.log
bash tools/r06_collect.sh
