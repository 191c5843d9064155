"""Timing of eigen_h (complex Hermitian) on the device API.  usage: gpu_herm_time.py N [m_forward=48] [reps=1]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eigenexa_amd import _lib
n = int(sys.argv[1]); mf = int(sys.argv[2]) if len(sys.argv) > 2 else 48; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = _lib.load(); _lib.check(lib.eigx_init(0), "init")
dev = torch.device("cuda:0"); torch.manual_seed(0)
B = torch.randn(n, n, dtype=torch.complex128, device=dev)
A = (B + B.conj().T) / 2          # Hermitian; at[j, i] = A(i, j) = conj(A[j, i]) -> column-major image is A.conj()... use A^T
tm = np.zeros(16)
for rep in range(reps + 1):
    at = A.T.contiguous().clone()  # at[j, i] = A(i, j)
    z = torch.zeros(n, n, dtype=torch.complex128, device=dev); w = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.eigx_h_dev(n, n, at.data_ptr(), n, w.data_ptr(), z.data_ptr(), n, mf, 128, b"A"), "eigen_h")
    dt = time.perf_counter() - t0
    lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
    Z = z.T                        # Z[:, k] = eigenvector k
    res = (torch.linalg.norm(A @ Z - Z * w[None, :]) / torch.linalg.norm(A)).item()
    orth = torch.linalg.norm(Z.conj().T @ Z - torch.eye(n, dtype=torch.complex128, device=dev)).item()
    print(f"n={n} mf={mf} rep {rep}: total {dt*1e3:.1f} ms  hrd {tm[1]*1e3:.1f}  dc {tm[2]*1e3:.1f}  hrbak {tm[3]*1e3:.1f}   "
          f"|AZ-ZW|/|A| = {res:.2e}  |Z^H Z - I| = {orth:.2e}", flush=True)
