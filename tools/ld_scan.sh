#!/bin/bash
# Lab script: (1) the real reduction at N=8192 on caller buffers with different leading dimensions, (2) eigen_h with
# different plane offsets and leading-dimension paddings.  Separate processes, two rounds.
cd "$(dirname "$0")/.."
for i in 1 2; do
  for lda in 8224 8256 8288 8320 8352 8448 8704 9216; do
    echo -n "reduce lda $lda: "
    EIGX_MF=64 EIGX_LDA=$lda timeout -k 10 200 python tools/gpu_reduce_time.py 8192 2 2 2>&1 | grep "rep [12]" | sed -e "s/(.*//" | tr "\n" " "
    echo
  done
done
for i in 1 2; do
  for sk in 896 1024 1040 1152 1280; do
    echo -n "herm skew $sk: "
    EIGX_H_SKEW=$sk timeout -k 10 300 python tools/gpu_herm_time.py 8192 48 1 2>&1 | grep "rep 1" | sed -e "s/|AZ.*//" -e "s/n=.*rep/rep/" | tr "\n" " "
    echo
  done
  for ex in 32 64 96 224; do
    echo -n "herm skew 1040 ld_extra $ex: "
    EIGX_H_SKEW=1040 EIGX_LD_EXTRA=$ex timeout -k 10 300 python tools/gpu_herm_time.py 8192 48 1 2>&1 | grep "rep 1" | sed -e "s/|AZ.*//" -e "s/n=.*rep/rep/" | tr "\n" " "
    echo
  done
done
