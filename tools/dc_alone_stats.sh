#!/bin/bash
# kernel stats of the band D&C alone (no back-transformation preparation beside it): tools/gpu_dc_check.py 8192 under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/dcalone
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 tools/gpu_dc_check.py 8192 > $out/check.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -n 1)
cut -c1-100 "$f" | head -n 3; grep -E "jacobi|secular|loewner|vectors|znext" "$f" | sed -E 's/\(anonymous namespace\):://g; s/\([^"]*\)"/"/' | cut -c1-160
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
tail -n 4 $out/check.log
