#!/bin/bash
# Round-4 final evidence (run from the repo root on the GPU box): kernel stats of the N=8192 bench and of one N=32768 solve
# under rocprofv3, the D&C timeline of the last N=8192 solve, the two PMC passes (HBM-side bytes) of the fused mat-vec.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04f
rm -rf $out && mkdir -p $out/pmc
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof8192 -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
echo "N=8192 under rocprof rc=$?"
f=$(find $out/prof8192 -name "*kernel_trace.csv" | head -n 1)
python3 tools/dc_timeline.py "$f" > $out/dc_timeline_n8192.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof32768 -o p -- python3 bench.py --size 32768 --steps 1 --warmup 1 --mf 256 --no-cpu-baseline --no-extra > $out/bench_n32768_under_rocprof.json 2> $out/bench_n32768_under_rocprof.err
echo "N=32768 under rocprof rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex symv_kernel --output-format csv -d $out/pmc/f -o f -- python3 tools/gpu_reduce_time.py 8192 2 0 > $out/pmc/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex symv_kernel --output-format csv -d $out/pmc/w -o w -- python3 tools/gpu_reduce_time.py 8192 2 0 > $out/pmc/w.log 2>&1
F=$(find $out/pmc/f -name "*counter_collection.csv" | head -1); W=$(find $out/pmc/w -name "*counter_collection.csv" | head -1)
python3 tools/symv_traffic.py 8192 "$F" "$W" $out/symv_traffic.json $out/symv_pmc_n8192.csv && cat $out/symv_traffic.json
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*.db" -delete
ls $out $out/prof8192 $out/prof32768 2>/dev/null | head -40
