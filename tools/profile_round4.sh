#!/bin/bash
# Round-4 evidence: kernel stats (CSV) of the N=8192 bench and of one N=32768 solve under rocprofv3, the loopback rehearsal of one
# rank of a 2 x 4 grid (plain and under the kernel trace).  Run from the repo root on the GPU box; the bench line itself
# (with the ~80-s CPU leg) is taken by a separate call.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof8192 -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
echo "N=8192 under rocprof rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof32768 -o p -- python3 bench.py --size 32768 --steps 1 --warmup 1 --mf 256 --no-cpu-baseline --no-extra > $out/bench_n32768_under_rocprof.json 2> $out/bench_n32768_under_rocprof.err
echo "N=32768 under rocprof rc=$?"
for fw in 0 1; do
  echo "== EIGX_FUSE_WAIT=$fw (0: role by role with wait kernels, 1: one launch per step)"
  EIGX_FUSE_WAIT=$fw python3 tools/mg_step_rehearsal.py 8 3 32768 2 128 2>&1 | grep -E "local block|rep 1"
done > $out/mg_step_rehearsal.log 2>&1
EIGX_FUSE_WAIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_mg -o p -- python3 tools/mg_step_rehearsal.py 8 3 32768 2 128 > $out/mg_rehearsal_under_rocprof.log 2>&1
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.db" -delete
cat $out/mg_step_rehearsal.log | cut -c1-200
ls $out $out/prof8192 $out/prof32768 $out/prof_mg 2>/dev/null | head -40
