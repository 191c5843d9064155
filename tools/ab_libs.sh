#!/bin/bash
# Lab script: A/B two builds of the library in alternating processes on ONE box (the box has two bandwidth levels that flip
# every tens of seconds -- DESIGN.md section 5 -- so sequential runs of one build each are not comparable).
#   1. build variant A, `cp eigenexa_amd/lib/libeigenexa_amd.so eigenexa_amd/lib/libeigx_a.so`; same for B -> libeigx_b.so
#   2. gpurun -- tools/ab_libs.sh a b 8192 2      (N, band; three rounds, two timed reductions each)
# Settings that eigx_tune can switch are better compared inside one process: EIGX_VARIANTS of tools/gpu_reduce_time.py.
cd "$(dirname "$0")/.."
A=$1; B=$2; N=${3:-8192}; BAND=${4:-2}
for i in 1 2 3; do
  for l in $A $B; do
    EIGX_LIB=eigenexa_amd/lib/libeigx_$l.so timeout -k 10 300 python tools/gpu_reduce_time.py $N $BAND 3 2>&1 | grep "rep [23]" |
      sed -e "s/ t128.*band=$BAND//" -e "s/(.*//" | tr "\n" " "
    echo
  done
done
