"""A/B of the D&C's side-stream threshold (eigx_tune key 15, value >= 2: merges larger than this run their secular / vector
kernels on the high-priority side stream under the previous product): N=8192 eigen_sx, D&C stage time per setting."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eigenexa_amd import _lib

lib = _lib.load()
_lib.check(lib.eigx_init(0), "eigx_init")
dev = torch.device("cuda:0")
n = 8192
torch.manual_seed(n)
R = torch.rand(n, n, dtype=torch.float64, device=dev)
A = R + R.T
lda = n + 512
for rnd in range(2):
    for side_min in (256, 512, 1024, 2048, 4096, 100000):
        lib.eigx_tune(15, side_min)
        ts = []
        for rep in range(3):
            a = torch.zeros(n, lda, dtype=torch.float64, device=dev); a[:, :n] = A.T
            z = torch.zeros(n, lda, dtype=torch.float64, device=dev); w = torch.zeros(n, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            _lib.check(lib.eigx_sx_dev(n, n, a.data_ptr(), lda, w.data_ptr(), z.data_ptr(), lda, 128, 128, b"A"), "solve")
            tm = np.zeros(16); lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
            ts.append(tm[2] * 1e3)
        print(f"round {rnd} side_min {side_min:6d}: D&C {min(ts):.2f} ms (min of {['%.2f' % t for t in ts]})", flush=True)
