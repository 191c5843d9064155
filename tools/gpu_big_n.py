"""Large-N check (BASELINE.json configs[4] size on one GPU): eigenvalues only (mode 'N': reduction + bisection) of
the N x N random symmetric matrix, validated through size-independent invariants: sum(w) = trace(A),
sum(w^2) = ||A||_F^2, sortedness.   usage: gpu_big_n.py N [route]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eigenexa_amd import _lib, layout

n = int(sys.argv[1]); route = sys.argv[2] if len(sys.argv) > 2 else "sx"
lib = _lib.load(); _lib.check(lib.eigx_init(0), "init")
dev = torch.device("cuda:0")
lda = n + 34
a = torch.empty(n, lda, dtype=torch.float64, device=dev)      # a[j, i] = A(i, j)
a[:, n:] = 0.0
tr = 0.0; fro2 = 0.0
ch = 4096
for c0 in range(0, n, ch):
    cols = np.arange(c0, min(n, c0 + ch))
    blk = layout.random_symmetric_torch(n, dev, rows=np.arange(n), cols=cols)   # (n, len(cols))
    a[c0:c0 + len(cols), :n] = blk.T
    fro2 += float((blk * blk).sum().item())
    tr += float(torch.diagonal(blk[c0:c0 + len(cols), :]).sum().item())
    del blk
w = torch.zeros(n, dtype=torch.float64, device=dev)
z = torch.zeros(8, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
fn = lib.eigx_sx_dev if route == "sx" else lib.eigx_s_dev
rc = fn(n, 0, a.data_ptr(), lda, w.data_ptr(), z.data_ptr(), lda, 128, 128, b"N")
dt = time.perf_counter() - t0
_lib.check(rc, "solve")
tm = np.zeros(16); lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
wh = w.cpu().numpy()
e1 = abs(wh.sum() - tr) / np.sqrt(fro2)
e2 = abs(np.sqrt((wh * wh).sum()) - np.sqrt(fro2)) / np.sqrt(fro2)
print(f"n={n} eigen_{route} mode N: {dt:.2f} s (reduction {tm[1]:.2f} s, bisection {tm[2]:.3f} s)  "
      f"|sum w - tr A|/|A|_F = {e1:.2e}  | |w|_2 - |A|_F | / |A|_F = {e2:.2e}  sorted={bool((np.diff(wh) >= 0).all())}", flush=True)
assert e1 < 1e-12 * np.sqrt(n) and e2 < 1e-12 and (np.diff(wh) >= 0).all()
