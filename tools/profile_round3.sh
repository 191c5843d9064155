#!/bin/bash
# Round-3 evidence: bench line, kernel stats (CSV) of the N=8192 bench and of one N=32768 solve.  Run from the repo root on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err
echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof8192 -o p -- python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/r03/bench_under_rocprof.json 2> gpurun_out/r03/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof32768 -o p -- python3 bench.py --size 32768 --steps 1 --warmup 1 --mf 256 --no-cpu-baseline --no-extra > gpurun_out/r03/bench_n32768_under_rocprof.json 2> gpurun_out/r03/bench_n32768_under_rocprof.err
find gpurun_out/r03 -name "*kernel_trace.csv" -delete
find gpurun_out/r03 -name "*.db" -delete
ls -la gpurun_out/r03 gpurun_out/r03/prof8192/* gpurun_out/r03/prof32768/* 2>/dev/null | head -30
tail -c 400 gpurun_out/r03/bench.json
