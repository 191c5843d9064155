#!/bin/bash
# Lab: re-sweep of the mat-vec switches on buffers with the channel-friendly leading dimension (EIGX_LDA), one buffer per sweep
cd "$(dirname "$0")/.."
f() { sed -e "s/t128.*band=[12]//" -e "s/(.*//"; }
echo "== N=8192 penta lda 8704 mf 64: unc threshold"; EIGX_MF=64 EIGX_LDA=8704 EIGX_VARIANTS="11=0;11=5000;11=9000;11=12000" timeout -k 10 200 python tools/gpu_reduce_time.py 8192 2 8 2>&1 | grep -v amdgpu | f
echo "== N=8192 penta lda 8704 mf 64: 128/256 tile switch"; EIGX_MF=64 EIGX_LDA=8704 EIGX_VARIANTS="3=3000;3=4500;3=6000;3=8192" timeout -k 10 200 python tools/gpu_reduce_time.py 8192 2 8 2>&1 | grep -v amdgpu | f
echo "== N=8192 penta lda 8704: panel width"; for mf in 32 48 64 96; do EIGX_MF=$mf EIGX_LDA=8704 timeout -k 10 200 python tools/gpu_reduce_time.py 8192 2 2 2>&1 | grep "rep 2" | f; done
echo "== N=16384 penta lda 16896 mf 128: unc threshold"; EIGX_MF=128 EIGX_LDA=16896 EIGX_VARIANTS="11=0;11=9000;11=12000;11=20000" timeout -k 10 300 python tools/gpu_reduce_time.py 16384 2 8 2>&1 | grep -v amdgpu | f
