"""GPU check of the band reduction (tri / penta) through the C-ABI against numpy (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eigenexa_amd import _lib

lib = _lib.load()
_lib.check(lib.eigx_init(0), "eigx_init")
dev = torch.device("cuda:0")


def band_matrix(d, e, band):
    n = len(d)
    T = np.diag(d)
    for b in range(1, min(band, n - 1) + 1):
        ee = e[b - 1][b:]
        T += np.diag(ee, b) + np.diag(ee, -b)
    return T


def run(n, band, m, seed=0, timing=False):
    rng = np.random.default_rng(seed)
    R = rng.random((n, n))
    A = R + R.T
    lda = n + (n & 1) + 2
    a = torch.zeros(n, lda, dtype=torch.float64, device=dev)  # a[j, i] = A(i, j) column-major
    a[:, :n] = torch.from_numpy(np.ascontiguousarray(A.T)).to(dev)
    # poison the strict lower triangle: it must never be read
    il = torch.tril_indices(n, n, -1, device=dev)
    a[il[1], il[0]] = float("nan")
    d = torch.zeros(n, dtype=torch.float64, device=dev)
    lde = n
    e = torch.zeros(band * lde, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.eigx_band_reduce_dev(n, a.data_ptr(), lda, d.data_ptr(), e.data_ptr(), lde, m, band)
    dt = time.perf_counter() - t0
    _lib.check(rc, "band_reduce")
    dn = d.cpu().numpy()
    en = e.cpu().numpy().reshape(band, lde)
    T = band_matrix(dn, en, band)
    w_band = np.linalg.eigvalsh(T)
    w_ref = np.linalg.eigvalsh(A)
    err = np.abs(w_band - w_ref).max() / np.abs(w_ref).max()
    print(f"n={n} band={band} m={m}: spectrum err {err:.2e}  time {dt*1e3:.1f} ms", flush=True)
    assert np.isfinite(err) and err < 1e-12 * max(1, n / 100), "band spectrum mismatch"
    return dt


if len(sys.argv) > 1:
    for band in (1, 2):
        run(int(sys.argv[1]), band, 128)
        run(int(sys.argv[1]), band, 128)
    sys.exit(0)
for band in (1, 2):
    for n, m in [(1, 8), (2, 8), (3, 8), (4, 8), (5, 8), (7, 4), (33, 8), (64, 16), (200, 32), (513, 48), (700, 128),
                 (1500, 64), (2049, 128)]:
        run(n, band, m)
for band in (1, 2):
    run(4096, band, 128)
    run(8192, band, 128)
    run(8192, band, 128)
print("REDUCE CHECK PASSED")
