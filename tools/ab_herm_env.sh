#!/bin/bash
# Lab script: eigen_h at N=8192 under different environment settings, alternating processes, three rounds.
#   gpurun -- tools/ab_herm_env.sh "EIGX_H_HEMV8_NT=0" "EIGX_H_HEMV8_NT=22" ...
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  for e in "$@"; do
    echo -n "$e: "
    env $e timeout -k 10 300 python tools/gpu_herm_time.py ${N:-8192} ${MF:-48} 1 2>&1 | grep "rep 1" | sed -e "s/|AZ.*//" -e "s/n=.*rep/rep/" | tr "\n" " "
    echo
  done
done
