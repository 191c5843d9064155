#!/bin/bash
# usage: tools/mg_stage_check.sh WORLD ROUTE DIMS n1,n2,...
cd "$(dirname "$0")/.."
W=$1; PORT=$((20000 + RANDOM % 20000)); pids=()
for ((r = 0; r < W; ++r)); do python tools/mg_stage_check.py $r $W $PORT $2 $3 $4 > /tmp/mgst_$r.log 2>&1 & pids+=($!); done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
grep -h "^n=" /tmp/mgst_0.log | cut -c1-400
[ $rc -eq 0 ] || { echo FAILED; for ((r = 0; r < W; ++r)); do tail -4 /tmp/mgst_$r.log | cut -c1-300; done; }
exit $rc
