#!/bin/bash
# Lab: multi-rank correctness subset + loopback rehearsal of one rank of a 2 x 4 grid at N = 32768 (variants through EIGX_TUNE)
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "multi_rank and (517 or 700 or 1111 or 4608 or 333 or folded)" 2>&1 | tail -2
for t in "" "3=20000" "11=1073741824" "12=1"; do
  echo "== EIGX_TUNE=$t"
  EIGX_TUNE=$t timeout -k 10 300 python tools/mg_step_rehearsal.py 8 3 32768 2 256 2>&1 | grep "rep 1"
done
echo "== m_forward = 128"
timeout -k 10 300 python tools/mg_step_rehearsal.py 8 3 32768 2 128 2>&1 | grep "rep 1"
