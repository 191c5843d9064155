#!/bin/bash
# kernel trace of a short N=8192 bench run; prints the D&C timeline of the last solve (tools/dc_timeline.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/dctl
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/bench.err
f=$(find $out/prof -name "*kernel_trace.csv" | head -n 1)
python3 tools/dc_timeline.py "$f" bt > $out/timeline.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
tail -n 2 $out/bench.json | cut -c1-400
