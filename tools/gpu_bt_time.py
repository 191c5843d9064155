"""Timing of the full solve stages for the back-transformation super-block factor q (eigx_tune key 2)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eigenexa_amd import _lib
n = int(sys.argv[1]); qs = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
lib = _lib.load(); _lib.check(lib.eigx_init(0), "init")
dev = torch.device("cuda:0"); torch.manual_seed(0)
lda = n + 34
R = torch.rand(n, lda, dtype=torch.float64, device=dev)
tm = np.zeros(16)
for rep in range(2):
    for q in qs:
        lib.eigx_tune(2, q)
        a = R.clone(); a[:, :n] = a[:, :n] + a[:, :n].T
        z = torch.zeros(n, lda, dtype=torch.float64, device=dev); w = torch.zeros(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        _lib.check(lib.eigx_sx_dev(n, n, a.data_ptr(), lda, w.data_ptr(), z.data_ptr(), lda, 128, 128, b"A"), "solve")
        lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
        print(f"n={n} q={q} rep {rep}: total {tm[0]*1e3:.1f} ms red {tm[1]*1e3:.1f} dc {tm[2]*1e3:.1f} bt {tm[3]*1e3:.1f}  "
              f"bt {2.0*n**3/tm[3]/1e12:.1f} TF", flush=True)
