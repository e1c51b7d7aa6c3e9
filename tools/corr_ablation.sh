#!/bin/bash
# Ablation builds of corr_disp_ring_kernel (each WRONG by design, timing only): tools/corr_ablation.sh build | run
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ABLS="1 2 4 8 17 10 14"
if [ "${1:-build}" = build ]; then
  python -m accflow_amd.build > /dev/null
  for m in $ABLS; do
    d=tools/bin/lib_cabl$m
    mkdir -p $d
    rm -rf $d/obj && cp -a accflow_amd/lib/obj $d/obj
    python -m accflow_amd.build --libdir=$d --unit-define=conv2d_direct.hip:ACCFLOW_CORR_ABL=$m | tail -1
  done
else
  for m in 0 $ABLS 0; do
    lib=accflow_amd/lib/libaccflow_hip.so; [ $m != 0 ] && lib=tools/bin/lib_cabl$m/libaccflow_hip.so
    ACCFLOW_HIP_LIB=$lib python tools/corr_gemm_bench.py 2>&1 | grep 60x128 | sed "s/^/ABL $m (1 no B reads, 2 no MFMA, 4 no DMA, 8 no store, 16 no A reads): /"
  done
fi
