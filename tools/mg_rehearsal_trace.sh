#!/bin/bash
# Lab: kernel trace (rocprofv3 --kernel-trace --stats) of the loopback rehearsal of one rank of a 2 x 4 grid at N = 32768:
# real per-kernel durations of the distributed step (the HIP-event brackets of mg_step_rehearsal.py add ~5 us each).
# usage: tools/mg_rehearsal_trace.sh <tag> [fuse_wait=1]       (run from the repo root on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r04}
export EIGX_FUSE_WAIT=${2:-1}
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 tools/mg_step_rehearsal.py 8 3 32768 2 128 > $out/rehearsal.log 2>&1
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.db" -delete
grep -E "rep 1" $out/rehearsal.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):7d} avg {float(r["AverageNs"])/1e3:8.2f} us total {float(r["TotalDurationNs"])/1e6:9.1f} ms')
PY
