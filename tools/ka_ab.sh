#!/bin/bash
# lab script: K_A workgroup cap A/B (eigx_tune key 7): 256 = one wave per SIMD (row-group loop), 100000 = one group per workgroup
cd "$(dirname "$0")/.."
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
e=d.get('extra',{})
print('N=8192', d['ms_per_step'], d['config']['stage_ms'], '| N=32768', e.get('seconds'), e.get('stage_ms'))"; }
for t in "7=100000" "7=256" "7=512" "7=128"; do
  echo "== EIGX_TUNE=$t"
  EIGX_TUNE=$t python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | show
done
for t in "7=100000" "7=256"; do
  echo "== EIGX_TUNE=$t (with the N=32768 extra solve)"
  EIGX_TUNE=$t python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | show
done
