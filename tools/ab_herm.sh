#!/bin/bash
# Lab script: A/B two builds of the library on eigen_h in alternating processes on ONE box (see tools/ab_libs.sh).
#   gpurun -- tools/ab_herm.sh a b [c,ENV=VAL,...]      (N, MF from the environment; a variant may carry environment settings)
cd "$(dirname "$0")/.."
N=${N:-8192}; MF=${MF:-48}
for i in 1 2 3; do
  for l in "$@"; do
    echo -n "$l: "
    lib=${l%%,*}; envs=""; [ "$lib" != "$l" ] && envs=$(echo "${l#*,}" | tr "," " ")
    env $envs EIGX_LIB=eigenexa_amd/lib/libeigx_$lib.so timeout -k 10 300 python tools/gpu_herm_time.py $N $MF 2 2>&1 | grep "rep [12]" |
      sed -e "s/|AZ.*//" -e "s/n=.*rep/rep/" | tr "\n" " "
    echo
  done
done
