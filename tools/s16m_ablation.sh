#!/bin/bash
# Experiment builds of the multi-source S16 kernel with one part of its loop removed each (results are wrong by design):
# which resource bounds the kernel?   usage: tools/s16m_ablation.sh   (then run tools/s16m_bench.py with ACCFLOW_HIP_LIB)
set -u
cd "$(dirname "$0")/.."
for v in NOA NODMA NOB NOMFMA; do
  d=tools/bin/lib_abl_$v
  mkdir -p $d/obj
  cp accflow_amd/lib/obj/*.o $d/obj/
  rm -f $d/obj/conv_s16m_v0.o $d/obj/conv_s16m_v1.o
  python -m accflow_amd.build --libdir=$PWD/$d -DS16M_ABL_$v=1 > $d/build.log 2>&1 &
  if [ "$v" = "NODMA" ]; then wait; fi
done
wait
ls -la tools/bin/lib_abl_*/libaccflow_hip.so
