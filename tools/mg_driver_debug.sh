#!/bin/bash
# lab script: the reference-format benchmark driver with 2 ranks on one GPU; prints rank 0's full report
cd "$(dirname "$0")/.."
printf '! N nvec bx by m t s e\n200 200 48 128 1 0 0 1\n131 131 32 64 1 2 1 1\n97 97 48 128 0 0 0 1\n-1 0 0 0 0 0 0 0\n' > /tmp/IN-mr
PORT=$((20000 + RANDOM % 20000))
for r in 0 1; do
  RANK=$r WORLD_SIZE=2 LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT EIGX_BENCH_BACKEND=gloo PYTHONPATH=$PWD \
    python -m eigenexa_amd.benchmark -f /tmp/IN-mr "$@" > /tmp/mr_$r.log 2>&1 &
done
wait
cat /tmp/mr_0.log
