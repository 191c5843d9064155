#!/bin/bash
# lab launcher of tools/mg_big_check.py: world processes on GPU 0 (at most 5: the box admits six GPU processes)
# usage: tools/mg_big_check.sh WORLD N ROUTE [PxxPy] [m_forward] [mode A|N]
cd "$(dirname "$0")/.."
W=$1; N=$2; RT=$3; DIMS=${4:--}; MF=${5:-128}; MODE=${6:-A}
PORT=$((20000 + RANDOM % 20000))
LOGD=${MGBIG_LOGDIR:-/tmp}
rm -f $LOGD/mgbig_*.log
pids=()
for ((r = 0; r < W; ++r)); do
  python tools/mg_big_check.py $r $W $PORT $N $RT $DIMS $MF $MODE > $LOGD/mgbig_$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
grep -h "^\[rank\|^OK\|Error\|error\|assert\|FAILED" $LOGD/mgbig_*.log | cut -c1-300
[ $rc -eq 0 ] || { echo "FAILED (tails follow)"; for ((r = 0; r < W; ++r)); do tail -5 $LOGD/mgbig_$r.log | cut -c1-300; done; }
exit $rc
