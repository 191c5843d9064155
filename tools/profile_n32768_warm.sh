#!/bin/bash
# Kernel trace of bench.py --size 32768 with one warm-up and two timed solves; the table is for the last (warm-card) solve
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03w/prof -o p -- python3 bench.py --size 32768 --steps 2 --warmup 1 --mf 256 --no-cpu-baseline --no-extra > gpurun_out/r03w/bench.json 2> gpurun_out/r03w/bench.err
t=$(find gpurun_out/r03w -name "*kernel_trace.csv" | head -1)
python tools/symv_last_solve.py "$t" 32768 3 > gpurun_out/r03w/symv_last_solve.txt 2>&1
cat gpurun_out/r03w/symv_last_solve.txt
find gpurun_out/r03w -name "*kernel_trace.csv" -delete
find gpurun_out/r03w -name "*.db" -delete
tail -c 700 gpurun_out/r03w/bench.json
