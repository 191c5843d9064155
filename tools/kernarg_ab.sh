#!/bin/bash
# lab script: bench.py with kernel arguments in host-visible memory (0) vs device memory (1)
cd "$(dirname "$0")/.."
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['stage_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
echo "== default env"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['stage_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
