#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench + PMC passes for the lookup kernel.
# Counter passes are separate runs with --kernel-trace only (no sys/hip/hsa trace domains).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${1:-r01}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-strict > $OUT/bench_under_rocprof.log 2>&1
# counter passes over the bench itself (real flows), pair groups on one stream so that every estimator lookup is B = 11
export ACCFLOW_STREAMS=1
PMC_CMD="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-strict"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $PMC_CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- $PMC_CMD > $OUT/pmc_l2.log 2>&1
unset ACCFLOW_STREAMS
python3 tools/lookup_bench.py --layout disp --flow 0 > $OUT/lookup_bench.log 2>&1
python3 tools/lookup_bench.py --layout disp --flow 0.1 --smooth 2 >> $OUT/lookup_bench.log 2>&1
python3 tools/lookup_bench.py --layout row >> $OUT/lookup_bench.log 2>&1
python3 tools/summarize_profiles.py $OUT
