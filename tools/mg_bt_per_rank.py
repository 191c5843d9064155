"""Lab: what the back-transformation costs ONE rank of a P-rank grid.  The multi-rank back-transformation gives every rank
ceil(nvec / P) eigenvector columns with all n rows and streams all reflectors past them, so a rank's
compute is that of a one-GPU solve with nvec / P eigenvectors (`nvec < N only trims the back-transform`):
this script runs exactly that on the idle card and prints the stage timer.  Not in it: the allgather of the reflector
groups (N^2 / 2 doubles received per rank over the 7 links) and the T-factor preparation's overlap with the D&C.
usage: mg_bt_per_rank.py N P [m_backward=128]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eigenexa_amd as ee
from eigenexa_amd import _lib, layout

n, P = int(sys.argv[1]), int(sys.argv[2])
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 128
lib = _lib.load()
ee.eigen_init()
dev = torch.device("cuda:0")
lda = ee.eigen_get_matdims(n)[0]
a0 = torch.empty(n, lda, dtype=torch.float64, device=dev)
a0[:, n:] = 0.0
for c0 in range(0, n, 4096):
    blk = layout.random_symmetric_torch(n, dev, rows=np.arange(n), cols=np.arange(c0, min(n, c0 + 4096)))
    a0[c0:c0 + blk.shape[1], :n] = blk.T
    del blk
w = torch.zeros(n, dtype=torch.float64, device=dev)
z = torch.empty(n, lda, dtype=torch.float64, device=dev)
tm = np.zeros(16)
for nvec in (n, (n + P - 1) // P, (n + P - 1) // P):
    a = a0.clone()
    torch.cuda.synchronize()
    _lib.check(lib.eigx_sx_dev(n, nvec, a.data_ptr(), lda, w.data_ptr(), z.data_ptr(), lda, 256, mb, b"A"), "eigen_sx")
    torch.cuda.synchronize()
    lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
    print(f"N={n} nvec={nvec}: total {tm[0]:.3f} s, reduction {tm[1]:.3f}, D&C {tm[2]:.3f}, back-transform {tm[3]*1e3:.1f} ms "
          f"({2.0 * nvec * n * n / max(tm[3], 1e-9) / 1e12:.1f} TFLOP/s)", flush=True)
    del a
