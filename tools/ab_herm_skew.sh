#!/bin/bash
# Lab script: eigen_h at N=8192 with different offsets (in doubles) between the real and the imaginary plane of A
cd "$(dirname "$0")/.."
for i in 1 2; do
  for sk in "$@"; do
    echo -n "skew $sk: "
    EIGX_H_SKEW=$sk timeout -k 10 300 python tools/gpu_herm_time.py 8192 48 1 2>&1 | grep "rep 1" |
      sed -e "s/|AZ.*//" -e "s/n=.*rep/rep/" | tr "\n" " "
    echo
  done
done
