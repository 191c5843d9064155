"""GPU check of the band divide-and-conquer through the C-ABI against numpy (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eigenexa_amd import _lib

lib = _lib.load()
_lib.check(lib.eigx_init(0), "eigx_init")
dev = torch.device("cuda:0")
eps = np.finfo(float).eps


def run(n, band, kind="rand", seed=0):
    rng = np.random.default_rng(seed)
    lde = n
    d = rng.standard_normal(n)
    e = np.zeros((band, lde))
    for b in range(1, band + 1):
        e[b - 1, b:] = rng.standard_normal(n - b) if n > b else []
    if kind == "glued":  # many near-identical blocks -> heavy deflation
        d = np.tile(rng.standard_normal(16), n // 16 + 1)[:n]
        for b in range(1, band + 1):
            e[b - 1, b:] = np.tile(rng.standard_normal(16), n // 16 + 1)[: n - b] * 1e-3
    if kind == "toeplitz":
        d[:] = 2.0
        e[0, 1:] = -1.0
        if band == 2:
            e[1, 2:] = 0.25
    T = np.diag(d)
    for b in range(1, min(band, n - 1) + 1):
        T += np.diag(e[b - 1, b:], b) + np.diag(e[b - 1, b:], -b)
    dd = torch.from_numpy(d).to(dev)
    ee = torch.from_numpy(e.reshape(-1).copy()).to(dev)
    ldz = n + 3
    z = torch.zeros(n, ldz, dtype=torch.float64, device=dev)
    w = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.eigx_band_dc_dev(n, n, dd.data_ptr(), ee.data_ptr(), lde, band, w.data_ptr(), z.data_ptr(), ldz)
    dt = time.perf_counter() - t0
    _lib.check(rc, "band_dc")
    wn = w.cpu().numpy()
    Z = z[:, :n].T.cpu().numpy()
    wr = np.linalg.eigvalsh(T)
    werr = np.abs(wn - wr).max() / max(np.abs(wr).max(), 1e-300)
    res = np.linalg.norm(T @ Z - Z * wn) / (n * eps * np.linalg.norm(T))
    orth = np.linalg.norm(Z.T @ Z - np.eye(n)) / (n * eps)
    print(f"n={n} band={band} {kind}: werr {werr:.2e} res {res:.3e} orth {orth:.3e} time {dt*1e3:.1f} ms", flush=True)
    assert werr < 1e-13 and res < 768 and orth < 8


sizes = [1, 2, 5, 33, 64, 65, 100, 129, 200, 513, 1000, 2048]
if len(sys.argv) > 1:
    sizes = [int(a) for a in sys.argv[1:]]
for band in (1, 2):
    for n in sizes:
        run(n, band)
    for n in [s for s in sizes if s >= 100][:3]:
        run(n, band, "glued")
        run(n, band, "toeplitz")
print("DC CHECK PASSED")
