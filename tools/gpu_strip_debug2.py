import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctypes as C
from eigenexa_amd import _lib, layout
lib = C.CDLL(_lib.LIB_PATH, mode=C.RTLD_GLOBAL)     # raw binding: also loads older builds of the library (EIGX_LIB)
for nm in ("eigx_sx_dev", "eigx_s_dev"):
    getattr(lib, nm).argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char]
assert lib.eigx_init(0) == 0
dev = torch.device("cuda:0")
EPS = np.finfo(np.float64).eps
CFG = {"tile": (1 << 30, 512, 24000), "h512": (700, 96, 1 << 30), "h1024": (700, 96, 0), "mixed": (700, 96, 2000)}
def run(route, name):
    n = 3000
    t11, t12, t13 = CFG[name]
    old = [lib.eigx_tune(11, t11), lib.eigx_tune(12, t12), lib.eigx_tune(13, t13)]
    try:
        A = layout.random_symmetric_torch(n, dev)
        a = torch.zeros(n, n + 34, dtype=torch.float64, device=dev)
        a[:, :n] = A.T
        z = torch.zeros(n, n + 34, dtype=torch.float64, device=dev)
        w = torch.zeros(n, dtype=torch.float64, device=dev)
        fn = lib.eigx_sx_dev if route == "sx" else lib.eigx_s_dev
        rc = fn(n, n, a.data_ptr(), n + 34, w.data_ptr(), z.data_ptr(), n + 34, 128, 128, b"A")
    finally:
        for key, v in zip((11, 12, 13), old): lib.eigx_tune(key, v)
    Z = z[:, :n].T
    anorm = torch.linalg.norm(A).item()
    res = torch.linalg.norm(A @ Z - Z * w[None, :]).item() / (n * EPS * anorm)
    orth = torch.linalg.norm(Z.T @ Z - torch.eye(n, dtype=torch.float64, device=dev)).item() / (n * EPS)
    wr = np.linalg.eigvalsh(A.cpu().numpy())
    werr = np.abs(w.cpu().numpy() - wr).max() / np.abs(wr).max()
    print(f"{route} {name}: rc {rc} residual {res:.3e} orth {orth:.3e} werr {werr:.1e} a@{a.data_ptr():x} z@{z.data_ptr():x}", flush=True)
for spec in sys.argv[1:]:
    r, c = spec.split(":")
    run(r, c)
