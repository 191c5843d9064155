"""GPU end-to-end check of eigx_sx_dev / eigx_s_dev against the reference's accuracy gates (run on GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eigenexa_amd import _lib

lib = _lib.load()
_lib.check(lib.eigx_init(0), "eigx_init")
dev = torch.device("cuda:0")
eps = np.finfo(float).eps


def frank(n):
    i = torch.arange(1, n + 1, dtype=torch.float64, device=dev)
    return torch.minimum(i[:, None], i[None, :])


def run(n, route, kind="rand", nvec=None, mode=b"A", reps=1):
    torch.manual_seed(n)
    if kind == "frank":
        A = frank(n)
    else:
        R = torch.rand(n, n, dtype=torch.float64, device=dev)
        A = R + R.T
    nvec = n if nvec is None else nvec
    lda = n + (n & 1) + 2
    fn = lib.eigx_sx_dev if route == "sx" else lib.eigx_s_dev
    best = 1e9
    for _ in range(reps):
        a = torch.zeros(n, lda, dtype=torch.float64, device=dev)
        a[:, :n] = A.T
        z = torch.zeros(n, lda, dtype=torch.float64, device=dev)
        w = torch.zeros(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = fn(n, nvec, a.data_ptr(), lda, w.data_ptr(), z.data_ptr(), lda, 128, 128, mode)
        dt = time.perf_counter() - t0
        _lib.check(rc, "solve")
        best = min(best, dt)
    tm = (np.zeros(16))
    import ctypes
    lib.eigx_get_timers(tm.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    Z = z[:nvec, :n].T  # n x nvec
    W = w[:nvec]
    anorm = torch.linalg.norm(A).item()
    res = torch.linalg.norm(A @ Z - Z * W[None, :]).item() / (n * eps * anorm)
    orth = torch.linalg.norm(Z.T @ Z - torch.eye(nvec, dtype=torch.float64, device=dev)).item() / (n * eps)
    if kind == "frank":
        k = np.arange(1, n + 1)
        lam = np.sort(1.0 / (2 * (1 - np.cos((2 * k - 1) * np.pi / (2 * n + 1)))))
        werr = np.abs((w.cpu().numpy() - lam) / lam).max()
    else:
        wr = torch.linalg.eigvalsh(A) if n <= 4096 else None
        werr = ((w - wr).abs().max() / wr.abs().max()).item() if wr is not None else float("nan")
    flops = a[0, 0].item()
    print(f"{route} n={n} {kind} nvec={nvec}: werr {werr:.2e} res {res:.3e} orth {orth:.3e} | {best*1e3:.1f} ms "
          f"(red {tm[1]*1e3:.1f} dc {tm[2]*1e3:.1f} bt {tm[3]*1e3:.1f}) {abs(flops)/best/1e9:.0f} GFLOP/s", flush=True)
    assert res < 768 and orth < 8
    if kind == "frank":
        assert werr < np.sqrt(eps)
    elif werr == werr:
        assert werr < 1e-12


sizes = [3, 4, 5, 7, 64, 200, 255, 256, 257, 1000, 1024]
if len(sys.argv) > 1:
    for n in [int(x) for x in sys.argv[1:]]:
        for route in ("sx", "s"):
            run(n, route, reps=2)
    sys.exit(0)
for route in ("sx", "s"):
    for n in sizes:
        run(n, route)
    run(1024, route, "frank")
    run(1000, route, nvec=50)
for route in ("sx", "s"):
    run(4096, route, reps=2)
    run(8192, route, reps=2)
print("SOLVE CHECK PASSED")
