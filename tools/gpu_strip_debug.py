"""lab: full solves with the strip form of the mat-vec forced on / off (residual and orthogonality per configuration);
every solve in a function of its own, so that the next one gets the same device addresses (as in the test suite)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eigenexa_amd import _lib, layout
lib = _lib.load(); _lib.check(lib.eigx_init(0), "init")
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
EPS = np.finfo(np.float64).eps
CFG = {"tile": (1 << 30, 512, 24000), "h512": (700, 96, 1 << 30), "h1024": (700, 96, 0), "mixed": (700, 96, 2000)}

def run(route, name):
    t11, t12, t13 = CFG[name]
    A = layout.random_symmetric_torch(n, dev)
    anorm = torch.linalg.norm(A).item()
    old = [lib.eigx_tune(11, t11), lib.eigx_tune(12, t12), lib.eigx_tune(13, t13)]
    a = torch.zeros(n, n + 34, dtype=torch.float64, device=dev); a[:, :n] = A.T
    z = torch.zeros(n, n + 34, dtype=torch.float64, device=dev)
    w = torch.zeros(n, dtype=torch.float64, device=dev)
    fn = lib.eigx_sx_dev if route == "sx" else lib.eigx_s_dev
    rc = fn(n, n, a.data_ptr(), n + 34, w.data_ptr(), z.data_ptr(), n + 34, 128, 128, b"A")
    for key, v in zip((11, 12, 13), old): lib.eigx_tune(key, v)
    Z = z[:, :n].T
    res = torch.linalg.norm(A @ Z - Z * w[None, :]).item() / (n * EPS * anorm)
    orth = torch.linalg.norm(Z.T @ Z - torch.eye(n, dtype=torch.float64, device=dev)).item() / (n * EPS)
    print(f"{route} {name}: rc {rc} residual {res:.3e} orth {orth:.3e}  a@{a.data_ptr():x}", flush=True)

for spec in sys.argv[2:]:
    r, c = spec.split(":")
    run(r, c)
