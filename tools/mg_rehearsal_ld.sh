#!/bin/bash
# Lab: loopback rehearsal of one rank of a 2 x 4 grid at N = 32768 with the old and the recommended leading dimension
cd "$(dirname "$0")/.."
for i in 1 2; do
  for nxm in old new; do
    echo "== nx $nxm, m_forward = 128"
    EIGX_NX=$nxm timeout -k 10 300 python tools/mg_step_rehearsal.py 8 3 32768 2 128 2>&1 | grep -E "local block|rep 1"
  done
done
