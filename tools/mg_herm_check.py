"""Lab tool: eigen_h on `world` processes sharing GPU 0 at a size the unit tests do not reach (sharded reduction by default,
EIGX_H_GATHER=1: the first, gathering version), timed per stage, checked on rank 0 through GPU matmuls of the GATHERED
eigenvectors: residual ||A Z - Z W||_F / (N eps ||A||_F) < 768, ||Z^H Z - I||_F / (N eps) < 8; w identical on every rank;
bytes the library holds.   argv: rank world port n [PxxPy]      launcher: tools/mg_herm_check.sh WORLD N [PxxPy]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import torch.distributed as dist

rank, world, port, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dims = tuple(int(v) for v in sys.argv[5].split("x")) if len(sys.argv) > 5 and "x" in sys.argv[5] else None
os.environ.setdefault("EIGX_COMM_TIMEOUT_S", "300")
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
import eigenexa_amd as ee
from eigenexa_amd import _lib, layout

lib = _lib.load()
ee.eigen_init(comm=True, device=0, dims=dims)
procs, Px, Py = ee.eigen_get_procs()
_, xi, yi = ee.eigen_get_id()
px, py = xi - 1, yi - 1
dev = torch.device("cuda", 0)
rows, cols = np.arange(px, n, Px), np.arange(py, n, Py)
nx, ny = ee.eigen_get_matdims(n)
loc = layout.random_hermitian(n, rows=rows, cols=cols)
a = torch.zeros(ny, nx, dtype=torch.complex128, device=dev)          # a[lj, li] = A(li, lj): column-major (nx, ny)
a[: len(cols), : len(rows)] = torch.from_numpy(np.ascontiguousarray(loc.T)).to(dev)
z = torch.zeros(ny, nx, dtype=torch.complex128, device=dev)
w = torch.zeros(n, dtype=torch.float64, device=dev)
for rep in range(2):
    a[: len(cols), : len(rows)] = torch.from_numpy(np.ascontiguousarray(loc.T)).to(dev)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    rc = lib.eigx_h_dev(n, n, a.data_ptr(), nx, w.data_ptr(), z.data_ptr(), nx, 48, 128, b"A")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert rc == 0, f"rank {rank}: status {rc}"
    tm = np.zeros(16)
    lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
    print(f"[rank {rank}] rep {rep} grid {Px}x{Py} n={n} eigen_h ({'gathered' if os.environ.get('EIGX_H_GATHER') == '1' else 'sharded'}): "
          f"{dt:.2f} s (reduction {tm[1]:.2f}, D&C {tm[2]:.2f}, back-transform {tm[3]:.2f}); held {lib.eigx_held_bytes() / 2**20:.0f} MiB "
          f"(hs. {lib.eigx_held_bytes_named(b'hs.') / 2**20:.0f}, hm. {lib.eigx_held_bytes_named(b'hm.') / 2**20:.0f})", flush=True)
wl = [torch.zeros(n, dtype=torch.float64) for _ in range(world)]
dist.all_gather(wl, w.cpu())
for q in range(world):
    assert torch.equal(wl[q], wl[0]), f"w differs between rank 0 and rank {q}"
bx, by = (n + Px - 1) // Px, (n + Py - 1) // Py
zl = torch.zeros(by, bx, dtype=torch.complex128)
zl[: len(cols), : len(rows)] = z[: len(cols), : len(rows)].cpu()
parts = [torch.zeros(by, bx, 2, dtype=torch.float64) for _ in range(world)]
dist.all_gather(parts, torch.view_as_real(zl).contiguous())
coords = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
dist.all_gather(coords, torch.tensor([px, py], dtype=torch.int64))
if rank == 0:
    Z = torch.zeros(n, n, dtype=torch.complex128, device=dev)
    for q in range(world):
        qx, qy = int(coords[q][0]), int(coords[q][1])
        blk = torch.view_as_complex(parts[q]).to(dev)                 # blk[lj, li]
        r_, c_ = torch.arange(qx, n, Px, device=dev), torch.arange(qy, n, Py, device=dev)
        Z[r_[:, None], c_[None, :]] = blk[: len(c_), : len(r_)].T
    A = torch.from_numpy(layout.random_hermitian(n)).to(dev)
    eps = np.finfo(float).eps
    res = (torch.linalg.norm(A @ Z - Z * w[None, :]) / (n * eps * torch.linalg.norm(A))).item()
    orth = (torch.linalg.norm(Z.conj().T @ Z - torch.eye(n, dtype=torch.complex128, device=dev)) / (n * eps)).item()
    print(f"[rank 0] residual metric {res:.3e} (< 768), unitarity {orth:.3e} (< 8)", flush=True)
    assert res < 768 and orth < 8
dist.barrier()
ee.eigen_free()
dist.destroy_process_group()
print(f"OK rank {rank}/{world}", flush=True)
