#!/bin/bash
# Lab: N=32768 band reduction with extra doubles on every INTERNAL leading dimension (panel, partial sums): EIGX_LD_EXTRA
cd "$(dirname "$0")/.."
for i in 1 2; do
  for ex in 0 32 96 160 288; do
    echo -n "ld_extra $ex: "
    EIGX_MF=256 EIGX_LDA=33280 EIGX_LD_EXTRA=$ex timeout -k 10 300 python tools/gpu_reduce_time.py 32768 2 1 2>&1 | grep "rep 1" | sed -e "s/(.*//" -e "s/default.*band=2//" | tr "\n" " "
    echo
  done
done
