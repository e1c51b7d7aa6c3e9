#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench (default run: 2 pair-group streams + the
# sequence pipeline's side stream; AND a single-stream, unpipelined run, so that per-launch averages of the B = 11 lookup / conv launches can be read from a tracked file), PMC passes
# for the lookup (HBM bytes) and the conv / correlation GEMM kernels (matrix-pipe busy, LDS issue stalls, clock), and a
# FETCH_SIZE calibration on a launch with a known byte count.  Counter passes are separate runs with --kernel-trace
# only (no sys/hip/hsa trace domains); the program comes directly after `--`.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-parity --no-strict --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 3 --warmup 1 > $OUT/bench_under_rocprof.log 2>&1
export ACCFLOW_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- $B --no-pipeline --steps 3 --warmup 1 > $OUT/bench_under_rocprof_1stream.log 2>&1
P="$B --no-pipeline --steps 1 --warmup 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $P > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $P > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- $P > $OUT/pmc_sq.log 2>&1
# configs[4] (AccFlow(GMA) 7x720x1280), one stream: which kernels carry the side measurement `other_configs` reports
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -- python3 bench.py --ofe gma --height 720 --width 1280 --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-parity --no-strict --no-extra > $OUT/bench_c5_under_rocprof.log 2>&1
unset ACCFLOW_STREAMS
# FETCH_SIZE calibration: a zero-flow lookup reads every byte of its 10x10 windows exactly once (1 600 B + 8 B coords per
# query pixel and launch, far beyond any cache), so counter / known bytes is the factor for this access pattern
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_cal -- python3 tools/lookup_bench.py --layout disp --flow 0 --reps 5 > $OUT/pmc_cal.log 2>&1
python3 tools/lookup_bench.py --layout disp --flow 0 > $OUT/lookup_bench.log 2>&1
python3 tools/lookup_bench.py --layout disp --flow 0.1 --smooth 2 >> $OUT/lookup_bench.log 2>&1
python3 tools/summarize_profiles.py $OUT $TAG
