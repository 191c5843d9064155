#!/bin/bash
# A/B of the CU mask of the back-transformation's preparation stream (EIGX_BT_CUMASK): duration of its Gram GEMM and of the
# D&C kernels that run beside it, D&C stage time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 11111111 01010101 0f0f0f0f 03030303; do
  out=gpurun_out/btmask_$m
  rm -rf $out && mkdir -p $out
  EIGX_BT_CUMASK=$m rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/bench.err
  f=$(find $out/prof -name "*kernel_trace.csv" | head -n 1)
  python3 tools/dc_timeline.py "$f" > $out/timeline.txt
  find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
  echo "== mask $m: $(head -n 1 $out/timeline.txt)"
  grep -E "gemm2_kernel<true, true|tbuild" $out/timeline.txt | head -n 4
  grep -E "secular_kernel<8>" $out/timeline.txt | head -n 6 | awk '{printf "%s ", $2} END {print " <- secular<8> us"}'
  python3 -c "
import json,sys
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('stage_ms', d['config'].get('stage_ms'), 'ms_per_step', d['ms_per_step'])"
done
