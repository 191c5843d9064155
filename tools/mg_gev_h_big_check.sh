#!/bin/bash
# lab: the multi-rank worker of the test suite at sizes beyond it (KMATH_EIGEN_GEV n=2000 on 2x2, eigen_h n=1500 on 1x3), ranks sharing GPU 0
cd "$GRAFT_REPO_ROOT"
PORT=$((20000 + RANDOM % 20000))
for r in 0 1 2 3; do EIGX_SELFTEST_ROUNDS=40 python tests/mg_worker.py $r 4 $PORT 2000 gev 0 2x2 > gpurun_out/gevbig_$r.log 2>&1 & done
wait
tail -n 2 gpurun_out/gevbig_*.log | cut -c1-200
PORT=$((20000 + RANDOM % 20000))
for r in 0 1 2; do EIGX_SELFTEST_ROUNDS=40 python tests/mg_worker.py $r 3 $PORT 1500 h 0 1x3 > gpurun_out/hbig_$r.log 2>&1 & done
wait
tail -n 2 gpurun_out/hbig_*.log | cut -c1-200
