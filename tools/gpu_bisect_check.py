"""GPU check of the Sturm-count bisection (eigx_band_bisect_dev) on band matrices against numpy."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eigenexa_amd import _lib

lib = _lib.load()
_lib.check(lib.eigx_init(0), "eigx_init")
dev = torch.device("cuda:0")


def band_dense(d, e, band):
    n = len(d)
    T = np.diag(d)
    for b in range(1, min(band, n - 1) + 1):
        T += np.diag(e[b - 1, b:n], b) + np.diag(e[b - 1, b:n], -b)
    return T


def run(n, band, kind="rand", seed=0, check=True):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal(n)
    e = np.zeros((band, n))
    for b in range(1, band + 1):
        e[b - 1, b:] = rng.standard_normal(max(n - b, 0))
    if kind == "zero_diag":
        d[:] = 0.0
    if kind == "sparse":
        e[:, ::3] = 0.0
        d[::2] = 0.0
    if kind == "const":
        d[:] = 2.0; e[0, 1:] = -1.0
        if band == 2: e[1, 2:] = 0.25
    if kind == "blocks":
        e[:, n // 2] = 0.0
        if band == 2: e[1, n // 2 + 1] = 0.0
    if kind == "graded":
        s = 10.0 ** np.linspace(0, -12, n)
        d *= s; e *= s[None, :]
    dt = torch.from_numpy(d).to(dev); et = torch.from_numpy(e.reshape(-1).copy()).to(dev)
    w = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.eigx_band_bisect_dev(n, dt.data_ptr(), et.data_ptr(), n, band, w.data_ptr()), "bisect")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    msg = f"n={n} band={band} {kind}: {ms:.2f} ms"
    if check:
        wr = np.linalg.eigvalsh(band_dense(d, e, band))
        err = np.abs(w.cpu().numpy() - wr).max() / max(1.0, np.abs(wr).max())
        msg += f"  max err {err:.2e}"
        print(msg, flush=True)
        assert err < 1e-13, "bisection mismatch"
    else:
        print(msg, flush=True)


for band in (1, 2):
    for n in (1, 2, 3, 4, 5, 7, 33, 100, 257, 1000):
        run(n, band)
    for kind in ("zero_diag", "sparse", "const", "blocks", "graded"):
        run(300, band, kind)
    run(2048, band)
for band in (1, 2):
    run(8192, band, check=False)
    run(8192, band, check=False)
    run(32768, band, check=False)
    run(65536, band, check=False)
print("BISECT CHECK PASSED")
