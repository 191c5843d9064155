#!/bin/bash
# lab script: trailing-update A/B through bench.py (main line N=8192 + extra N=32768): C tiles streamed or not, m = 128 / 256
cd "$(dirname "$0")/.."
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
e=d.get('extra',{})
print('N=8192', d['ms_per_step'], d['config']['stage_ms'], 'K1', d['roofline_trailing_update']['achieved'], '| N=32768', e.get('seconds'), e.get('stage_ms'), 'K1', e.get('roofline_trailing_update',{}).get('achieved'), 'symv', e.get('roofline',{}).get('achieved'))"; }
for t in "6=0" "6=1"; do for mf in 128 256; do
  echo "== EIGX_TUNE=$t extra-mf=$mf"
  EIGX_TUNE=$t python bench.py --steps 2 --warmup 1 --no-cpu-baseline --extra-mf $mf 2>&1 | tail -1 | show
done; done
