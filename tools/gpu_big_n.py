"""Large-N check (BASELINE.json configs[4] size on one GPU): eigenvalues only (mode 'N': reduction + bisection) of
the N x N random symmetric matrix, validated through size-independent invariants: sum(w) = trace(A),
sum(w^2) = ||A||_F^2, sortedness.   usage: gpu_big_n.py N [route] [mode]
mode A additionally computes all eigenvectors and checks residual and orthogonality on a random sample of 512 of them
(A is regenerated chunk by chunk for the check, so the whole run fits one 288-GB GPU at N = 65536)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eigenexa_amd import _lib, layout

n = int(sys.argv[1]); route = sys.argv[2] if len(sys.argv) > 2 else "sx"
mode = sys.argv[3] if len(sys.argv) > 3 else "N"
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
z = torch.zeros(8, dtype=torch.float64, device=dev) if mode == "N" else torch.empty(n, lda, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
fn = lib.eigx_sx_dev if route == "sx" else lib.eigx_s_dev
rc = fn(n, 0 if mode == "N" else n, a.data_ptr(), lda, w.data_ptr(), z.data_ptr(), lda, 128, 128, mode.encode())
dt = time.perf_counter() - t0
_lib.check(rc, "solve")
tm = np.zeros(16); lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
wh = w.cpu().numpy()
e1 = abs(wh.sum() - tr) / np.sqrt(fro2)
e2 = abs(np.sqrt((wh * wh).sum()) - np.sqrt(fro2)) / np.sqrt(fro2)
print(f"n={n} eigen_{route} mode {mode}: {dt:.2f} s (reduction {tm[1]:.2f} s, D&C / bisection {tm[2]:.3f} s, back-transform {tm[3]:.2f} s)  "
      f"|sum w - tr A|/|A|_F = {e1:.2e}  | |w|_2 - |A|_F | / |A|_F = {e2:.2e}  sorted={bool((np.diff(wh) >= 0).all())}", flush=True)
assert e1 < 1e-12 * np.sqrt(n) and e2 < 1e-12 and (np.diff(wh) >= 0).all()
if mode != "N":
    del a
    torch.cuda.empty_cache()
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    idx = torch.randperm(n, generator=g)[:512].sort().values.to(dev)
    Zs = z[idx, :n].T.contiguous()                      # (n, 512) sampled eigenvectors
    ws = w[idx]
    R = torch.zeros_like(Zs)
    for c0 in range(0, n, ch):
        cols = np.arange(c0, min(n, c0 + ch))
        blk = layout.random_symmetric_torch(n, dev, rows=np.arange(n), cols=cols)
        R += blk @ Zs[c0:c0 + len(cols), :]
        del blk
    R -= Zs * ws[None, :]
    res = float(R.norm(dim=0).max().item()) / np.sqrt(fro2)
    G = z[:, :n] @ Zs                                    # (n, 512): Z^T Zs must be the sampled columns of I
    G[idx, torch.arange(512, device=dev)] -= 1.0
    orth = float(G.abs().max().item())
    print(f"  sampled 512 eigenpairs: max_k |A z_k - w_k z_k| / |A|_F = {res:.2e} (gate 1e-12 N = {1e-12 * n:.1e});  "
          f"max |Z^T z_k - e_k| = {orth:.2e};  peak HBM {torch.cuda.max_memory_allocated() / 2**30:.0f} GiB (torch side)", flush=True)
    assert res < 1e-12 * n and orth < 1e-10
