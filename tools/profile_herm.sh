#!/bin/bash
# Kernel stats of one eigen_h solve (N=8192) under rocprofv3.  Run from the repo root on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/herm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/herm/prof -o p -- python3 tools/gpu_herm_time.py ${1:-8192} ${2:-48} 1 > gpurun_out/herm/run.log 2>&1
t=$(find gpurun_out/herm -name "*kernel_trace.csv" | head -1)
python tools/herm_trace_table.py "$t" ${1:-8192} > gpurun_out/herm/by_size.txt 2>&1
cat gpurun_out/herm/by_size.txt
find gpurun_out/herm -name "*kernel_trace.csv" -delete
find gpurun_out/herm -name "*.db" -delete
cat gpurun_out/herm/run.log | tail -3
f=$(find gpurun_out/herm -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -c1-200
