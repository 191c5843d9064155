#!/bin/bash
# What would ka_kernel gain from fewer partial-sum slots at large L (VERDICT r3 #3, row-walk super-tiles)?  Upper bound by an
# existing switch: 512-tiles from L = 12000 on HALVE every row's slots (a 4-tile row walk removes 37 %).  N=32768 reduction,
# m_forward = 256, kernel stats of both settings.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t256 in 40000 12000; do
  out=gpurun_out/kaslots_$t256
  rm -rf $out && mkdir -p $out
  EIGX_T256=$t256 EIGX_MF=256 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 tools/gpu_reduce_time.py 32768 2 1 > $out/run.log 2>&1
  echo "== 256-tiles up to L = $t256 (beyond: 512-tiles)"
  grep -E "TFLOP" $out/run.log | cut -c1-200
  f=$(find $out/prof -name "*kernel_stats.csv" | head -n 1)
  grep -E "ka_kernel|symv_kernel" "$f" | sed -E 's/\(anonymous namespace\):://g; s/\(eigx::RedArgs[^"]*"/"/' | cut -c1-150
  find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
done
