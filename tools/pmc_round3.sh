#!/bin/bash
# Round-3: panel-width sweep of the N=8192 reduction, and the two PMC passes (HBM-side bytes) of the fused mat-vec
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mf in 48 64 96 128; do EIGX_MF=$mf timeout -k 10 120 python tools/gpu_reduce_time.py 8192 2 3 2>&1 | grep "rep [23]" | sed -e "s/ t128.*nt=-//" -e "s/(.*//"; done
mkdir -p gpurun_out/r03/pmc
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex symv_kernel --output-format csv -d gpurun_out/r03/pmc/f -o f -- python3 tools/gpu_reduce_time.py 8192 2 0 > gpurun_out/r03/pmc/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex symv_kernel --output-format csv -d gpurun_out/r03/pmc/w -o w -- python3 tools/gpu_reduce_time.py 8192 2 0 > gpurun_out/r03/pmc/w.log 2>&1
F=$(find gpurun_out/r03/pmc/f -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r03/pmc/w -name "*counter_collection.csv" | head -1)
echo "F=$F W=$W"
python tools/symv_traffic.py 8192 "$F" "$W" gpurun_out/r03/symv_traffic.json gpurun_out/r03/symv_pmc_n8192.csv && cat gpurun_out/r03/symv_traffic.json
find gpurun_out/r03/pmc -name "*kernel_trace.csv" -delete; find gpurun_out/r03/pmc -name "*counter_collection.csv" -delete
