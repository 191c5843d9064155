timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "load_forms or default_stream or symv_tile_sizes or tridiagonal_matches or pentadiagonal_is or ka_load_batch" 2>&1 | tail -3
f() { sed -e "s/t128.*band=[12]//" -e "s/(.*//"; }
echo "== N=8192 penta: unc threshold"; EIGX_VARIANTS="11=0;11=3000;11=5000;11=9000" timeout -k 10 200 python tools/gpu_reduce_time.py 8192 2 8 2>&1 | grep -v amdgpu | f
echo "== N=8192 penta: 128/256 tile switch (unc everywhere)"; EIGX_VARIANTS="3=3000;3=4500;3=6000;3=8192" timeout -k 10 200 python tools/gpu_reduce_time.py 8192 2 8 2>&1 | grep -v amdgpu | f
echo "== N=8192 tri: unc threshold"; EIGX_VARIANTS="11=0;11=5000;11=9000" timeout -k 10 200 python tools/gpu_reduce_time.py 8192 1 6 2>&1 | grep -v amdgpu | f
echo "== N=16384 penta: unc threshold"; EIGX_VARIANTS="11=0;11=9000;11=12000;11=20000" timeout -k 10 300 python tools/gpu_reduce_time.py 16384 2 8 2>&1 | grep -v amdgpu | f
echo "== N=32768 penta mf=256: K_A workgroups"; EIGX_MF=256 EIGX_VARIANTS="7=256;7=512;7=1024;7=128" timeout -k 10 400 python tools/gpu_reduce_time.py 32768 2 8 2>&1 | grep -v amdgpu | f
