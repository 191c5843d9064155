"""Lab tool: which stage of the multi-rank solver goes wrong at a given size?  `world` processes on GPU 0.
argv: rank world port route(sx|s) dims n1,n2,...
per size: (1) eigenvalues only (mode N: reduction + D&C without vectors) against torch.linalg.eigvalsh(A); (2) full
solve: eigenvalues, residual / orthogonality and the list of bad columns from rank 0's gathered Z."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import torch.distributed as dist

rank, world, port, route = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dims = tuple(int(v) for v in sys.argv[5].split("x")) if "x" in sys.argv[5] else None
sizes = [int(v) for v in sys.argv[6].split(",")]
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
band = 2 if route == "sx" else 1
for n in sizes:
    rows = np.arange(px, n, Px); cols = np.arange(py, n, Py)
    nx, ny = ee.eigen_get_matdims(n)
    loc = layout.random_symmetric_torch(n, dev, rows=rows, cols=cols)
    def fresh():
        a = torch.zeros(ny, nx, dtype=torch.float64, device=dev)
        a[: len(cols), : len(rows)] = loc.T
        return a
    wref = None
    if rank == 0:
        wref = torch.linalg.eigvalsh(layout.random_symmetric_torch(n, dev))
    msg = f"n={n} {Px}x{Py} {route}:"
    # (1) eigenvalues only
    fn = lib.eigx_sx_dev if route == "sx" else lib.eigx_s_dev
    a = fresh()
    w = torch.zeros(n, dtype=torch.float64, device=dev); z = torch.zeros(ny, nx, dtype=torch.float64, device=dev)
    rc = fn(n, n, a.data_ptr(), nx, w.data_ptr(), z.data_ptr(), nx, 128, 128, b"N")
    torch.cuda.synchronize()
    if rank == 0:
        msg += f" | mode N rc={rc} w err {(w - wref).abs().max().item() / wref.abs().max().item():.2e}"
    # (2) full
    a = fresh()
    rc = fn(n, n, a.data_ptr(), nx, w.data_ptr(), z.data_ptr(), nx, 128, 128, b"A")
    torch.cuda.synchronize()
    mr, mc = (n + Px - 1) // Px, (n + Py - 1) // Py
    zl = torch.zeros(mc, mr, dtype=torch.float64)
    zl[: len(cols), : len(rows)] = z[: len(cols), : len(rows)].cpu()
    if rank == 0:
        msg += f" | mode A rc={rc} w err {(w - wref).abs().max().item() / wref.abs().max().item():.2e}"
        Z = torch.zeros(n, n, dtype=torch.float64, device=dev)
        for q in range(world):
            blk = zl if q == 0 else torch.zeros(mc, mr, dtype=torch.float64)
            cq = torch.tensor([px, py], dtype=torch.int64)
            if q: dist.recv(blk, src=q); dist.recv(cq, src=q)
            r_ = torch.arange(int(cq[0]), n, Px, device=dev); c_ = torch.arange(int(cq[1]), n, Py, device=dev)
            Z[c_[:, None], r_[None, :]] = blk[: len(c_), : len(r_)].to(dev)
        A = layout.random_symmetric_torch(n, dev); Zm = Z.T
        eps = np.finfo(float).eps
        res = torch.linalg.norm(A @ Zm - Zm * w[None, :]).item() / (n * eps * torch.linalg.norm(A).item())
        orth = torch.linalg.norm(Zm.T @ Zm - torch.eye(n, dtype=torch.float64, device=dev)).item() / (n * eps)
        # which columns are bad?
        colres = torch.linalg.norm(A @ Zm - Zm * w[None, :], dim=0)
        bad = (colres > 1e-8).nonzero().flatten()
        msg += f" res {res:.3g} orth {orth:.3g} bad columns {bad.numel()} {bad[:6].tolist()}..{bad[-3:].tolist() if bad.numel() else ''}"
        print(msg, flush=True)
        del A, Z, Zm
    else:
        dist.send(zl, dst=0); dist.send(torch.tensor([px, py], dtype=torch.int64), dst=0)
    dist.barrier()
ee.eigen_free()
dist.destroy_process_group()
