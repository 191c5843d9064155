"""Lab: what makes the N=32768 reduction slower inside bench.py (5.1 s) than in a fresh process (4.68 s)?  Mimics bench.py's
allocation pattern step by step.  usage: gpu_bench_mimic.py N [variant]   variant bits: 1 = separate buffer per solve,
2 = keep the generated matrix (transposed copy) alive, 4 = ee.eigen_init() instead of eigx_init, 8 = eigx_profile(8),
16 = arrays of [ny, nx] from eigen_get_matdims"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import eigenexa_amd as ee
from eigenexa_amd import _lib, layout
n = int(sys.argv[1]); var = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
if var & 4: ee.eigen_init()
else: _lib.check(lib.eigx_init(0), "init")
dev = torch.device("cuda:0")
nx, ny = ee.eigen_get_matdims(n)
if not (var & 16): ny = n
loc = torch.empty(n, n, dtype=torch.float64, device=dev)
for c0 in range(0, n, 4096):
    blk = layout.random_symmetric_torch(n, dev, rows=np.arange(n), cols=np.arange(c0, c0 + 4096))
    loc[:, c0:c0 + 4096] = blk
    del blk
A_T = loc.T.contiguous(); del loc
bufs = []
for _ in range(2 if var & 1 else 1):
    a = torch.zeros(ny, nx, dtype=torch.float64, device=dev); a[:n, :n] = A_T; bufs.append(a)
if not (var & 2): 
    keep = A_T.clone() if not (var & 1) else None   # single-buffer mode needs a pristine copy
    if var & 1: del A_T; torch.cuda.empty_cache()
z = torch.zeros(ny, nx, dtype=torch.float64, device=dev); w = torch.zeros(n, dtype=torch.float64, device=dev)
tm = np.zeros(16)
for rep in range(2):
    a = bufs[rep if var & 1 else 0]
    if not (var & 1) and rep: a[:n, :n] = A_T if (var & 2) else keep
    if var & 8: lib.eigx_profile(8)
    torch.cuda.synchronize()
    _lib.check(lib.eigx_sx_dev(n, n, a.data_ptr(), nx, w.data_ptr(), z.data_ptr(), nx, 256, 128, b"A"), "sx")
    lib.eigx_profile(0)
    lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
    print(f"variant {var} rep {rep}: total {tm[0]*1e3:.1f} ms, reduction {tm[1]*1e3:.1f}, dc {tm[2]*1e3:.1f}, bt {tm[3]*1e3:.1f}", flush=True)
