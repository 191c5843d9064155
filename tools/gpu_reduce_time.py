"""Timing only: band reduction (eigx_band_reduce_dev) of a random symmetric matrix generated on the GPU.
usage: gpu_reduce_time.py N [band=2] [reps=1]   (library selectable with EIGX_LIB for A/B runs; EIGX_MF = panel width)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigenexa_amd import _lib

n = int(sys.argv[1]); band = int(sys.argv[2]) if len(sys.argv) > 2 else 2; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = _lib.load()
_lib.check(lib.eigx_init(0), "eigx_init")
if os.environ.get("EIGX_T128"): lib.eigx_tune(3, int(os.environ["EIGX_T128"]))
if os.environ.get("EIGX_T256"): lib.eigx_tune(4, int(os.environ["EIGX_T256"]))
if os.environ.get("EIGX_NT"): lib.eigx_tune(5, int(os.environ["EIGX_NT"]))
dev = torch.device("cuda:0")
torch.manual_seed(0)
lda = n + (n & 1) + 2 + 30   # even, not a multiple of a large power of two
R = torch.rand(n, lda, dtype=torch.float64, device=dev)
d = torch.zeros(n, dtype=torch.float64, device=dev)
e = torch.zeros(band * n, dtype=torch.float64, device=dev)
for rep in range(reps + 1):
    a = R.clone()
    a[:, :n] = a[:, :n] + a[:, :n].T
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.eigx_band_reduce_dev(n, a.data_ptr(), lda, d.data_ptr(), e.data_ptr(), n, int(os.environ.get('EIGX_MF', '128')), band), "reduce")
    dt = time.perf_counter() - t0
    print(f"{os.environ.get('EIGX_LIB', 'default')[-24:]} t128={os.environ.get('EIGX_T128','-')} t256={os.environ.get('EIGX_T256','-')} nt={os.environ.get('EIGX_NT','-')} mf={os.environ.get('EIGX_MF','128')} n={n} band={band} rep {rep}: {dt*1e3:.1f} ms  "
          f"({4.0/3.0*n**3/dt/1e12:.2f} TFLOP/s of 4/3 n^3; d[0]={d[0].item():.6f})", flush=True)
