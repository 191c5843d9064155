"""Lab tool: the multi-rank solver at BASELINE-like sizes with `world` processes sharing GPU 0 (peer-window transport, the
device code that runs over xGMI on a node) and a full check on rank 0: residual ||A Z - Z W||_F / (N eps ||A||_F) < 768,
orthogonality ||Z^T Z - I||_F / (N eps) < 8 through GPU matmuls, w bit-identical on every rank.
argv: rank world port n route(sx|s) [PxxPy] [m_forward] [mode A|N]   (mode N: eigenvalues only, checked through
sum(w) = tr(A), sum(w^2) = ||A||_F^2 and sortedness -- the invariants of test_baseline_config_n65536_eigenvalues_only)
launcher: tools/mg_big_check.sh"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import torch.distributed as dist

rank, world, port, n, route = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
dims = tuple(int(v) for v in sys.argv[6].split("x")) if len(sys.argv) > 6 and "x" in sys.argv[6] else None
mf = int(sys.argv[7]) if len(sys.argv) > 7 else 128
mode = sys.argv[8] if len(sys.argv) > 8 else "A"
os.environ.setdefault("EIGX_COMM_TIMEOUT_S", "300")
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
import eigenexa_amd as ee
from eigenexa_amd import _lib, api, layout

lib = _lib.load()
ee.eigen_init(comm=True, device=0, dims=dims)
procs, Px, Py = ee.eigen_get_procs()
_, xi, yi = ee.eigen_get_id()
px, py = xi - 1, yi - 1
dev = torch.device("cuda", 0)
rows = np.arange(px, n, Px)
cols = np.arange(py, n, Py)
nx, ny = ee.eigen_get_matdims(n)
loc = layout.random_symmetric_torch(n, dev, rows=rows, cols=cols)
a = torch.zeros(ny, nx, dtype=torch.float64, device=dev)
a[: len(cols), : len(rows)] = loc.T
inv = torch.zeros(2, dtype=torch.float64)
rr_, cc_ = torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)
inv[0] = (loc * (rr_[:, None] == cc_[None, :])).sum().item()      # my share of tr(A)
inv[1] = (loc * loc).sum().item()                                 # ... and of ||A||_F^2
dist.all_reduce(inv)
del loc
z = torch.zeros(ny, nx, dtype=torch.float64, device=dev) if mode == "A" else torch.zeros(8, dtype=torch.float64, device=dev)
w = torch.zeros(n, dtype=torch.float64, device=dev)
fn = lib.eigx_sx_dev if route == "sx" else lib.eigx_s_dev
torch.cuda.synchronize()
dist.barrier()
print(f"[rank {rank}] solving n={n} on {Px}x{Py} ...", flush=True)
t0 = time.perf_counter()
rc = fn(n, n if mode == "A" else 0, a.data_ptr(), nx, w.data_ptr(), z.data_ptr(), nx, mf, 128, mode.encode())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert rc == 0, f"rank {rank}: status {rc}"
tm = np.zeros(16)
lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
held = lib.eigx_held_bytes() if hasattr(lib, "eigx_held_bytes") else -1
print(f"[rank {rank}] grid {Px}x{Py} n={n} {route} mf={mf}: {dt:.2f} s (reduction {tm[1]:.2f}, D&C {tm[2]:.2f}, back-transform {tm[3]:.2f}, "
      f"comm {tm[4]:.3f}); held {held / 2**30:.2f} GiB", flush=True)

# w identical on every rank
wl = [torch.zeros(n, dtype=torch.float64) for _ in range(world)]
dist.all_gather(wl, w.cpu())
for q in range(world):
    assert torch.equal(wl[q], wl[0]), f"w differs between rank 0 and rank {q}"

anorm_ = float(inv[1].sqrt())
tr_err = abs(float(w.sum().item()) - float(inv[0])) / anorm_
fro_err = abs(float(torch.linalg.norm(w).item()) - anorm_) / anorm_
srt = bool((w[1:] >= w[:-1]).all().item())
if rank == 0:
    print(f"[rank 0] trace error / ||A|| {tr_err:.2e}, Frobenius error / ||A|| {fro_err:.2e}, ascending {srt} (w[0] {float(w[0]):.6f}, "
          f"w[-1] {float(w[-1]):.6f})", flush=True)
assert tr_err < 1e-12 and fro_err < 1e-12 and srt
if mode != "A":
    dist.barrier()
    ee.eigen_free()
    dist.destroy_process_group()
    print(f"OK rank {rank}/{world}", flush=True)
    sys.exit(0)

# gather the blocks of z on rank 0 (one block at a time: bounded host memory)
mr, mc = (n + Px - 1) // Px, (n + Py - 1) // Py
zl = torch.zeros(mc, mr, dtype=torch.float64)
zl[: len(cols), : len(rows)] = z[: len(cols), : len(rows)].cpu()
if rank == 0:
    Z = torch.zeros(n, n, dtype=torch.float64, device=dev)   # Z[j, i] = z(i, j): column-major global matrix
    for q in range(world):
        if q == 0:
            blk = zl
        else:
            blk = torch.zeros(mc, mr, dtype=torch.float64)
            dist.recv(blk, src=q)
        # rank q's grid coordinates
        cq = torch.zeros(2, dtype=torch.int64)
        if q == 0:
            cq[0], cq[1] = px, py
        else:
            dist.recv(cq, src=q)
        qx, qy = int(cq[0]), int(cq[1])
        r_ = torch.arange(qx, n, Px, device=dev)
        c_ = torch.arange(qy, n, Py, device=dev)
        Z[c_[:, None], r_[None, :]] = blk[: len(c_), : len(r_)].to(dev)
    del a, z
    torch.cuda.empty_cache()
    A = layout.random_symmetric_torch(n, dev)
    Zm = Z.T                                           # (row i, column j) view
    wv = wl[0].to(dev)
    eps = np.finfo(np.float64).eps
    anorm = torch.linalg.norm(A).item()
    res = torch.linalg.norm(A @ Zm - Zm * wv[None, :]).item() / (n * eps * anorm)
    orth = torch.linalg.norm(Zm.T @ Zm - torch.eye(n, dtype=torch.float64, device=dev)).item() / (n * eps)
    tr_err = abs(wv.sum().item() - torch.trace(A).item()) / anorm
    ok = res < 768 and orth < 8
    print(f"[rank 0] residual metric {res:.4f} (< 768), orthogonality metric {orth:.4f} (< 8), trace error / ||A|| {tr_err:.2e}: "
          f"{'OK' if ok else 'FAILED'}", flush=True)
    assert ok
else:
    dist.send(zl, dst=0)
    dist.send(torch.tensor([px, py], dtype=torch.int64), dst=0)
dist.barrier()
ee.eigen_free()
dist.destroy_process_group()
print(f"OK rank {rank}/{world}", flush=True)
