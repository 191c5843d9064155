#!/bin/bash
# lab launcher of tools/mg_herm_check.py: world processes on GPU 0      usage: tools/mg_herm_check.sh WORLD N [PxxPy]
cd "$(dirname "$0")/.."
W=$1; N=$2; DIMS=${3:--}
PORT=$((20000 + RANDOM % 20000))
LOGD=${MGBIG_LOGDIR:-/tmp}
rm -f $LOGD/mgherm_*.log
pids=()
for ((r = 0; r < W; ++r)); do
  python tools/mg_herm_check.py $r $W $PORT $N $DIMS > $LOGD/mgherm_$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
grep -h "^\[rank\|^OK\|Error\|error\|assert\|FAILED" $LOGD/mgherm_*.log | cut -c1-300
[ $rc -eq 0 ] || { echo "FAILED (tails follow)"; for ((r = 0; r < W; ++r)); do tail -5 $LOGD/mgherm_$r.log | cut -c1-300; done; }
exit $rc
