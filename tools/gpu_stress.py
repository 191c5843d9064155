"""Randomised sequence of solves through the Python mirror (host API): sizes, routes (sx / s / h), modes and panel widths
change from call to call, with eigen_free / eigen_init cycles in between -- catches state that leaks between solves
(pooled workspace, prepared back-transformation plans, zero-padding assumptions).  usage: gpu_stress.py [ncalls] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eigenexa_amd as ee
from eigenexa_amd import api

ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
EPS = np.finfo(float).eps
ee.eigen_init()
worst = [0.0, 0.0, 0.0]
for it in range(ncalls):
    if it % 37 == 36:
        ee.eigen_free(); ee.eigen_init()
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 65, 127, 128, 129, 200, 257, 300, 513, 640, 1025]))
    route = str(rng.choice(["sx", "s", "h"]))
    mode = str(rng.choice(["A", "A", "A", "N", "X"]))
    nvec = n if rng.random() < 0.7 else int(rng.integers(1, n + 1))
    mf = int(rng.choice([8, 32, 48, 128])); mb = int(rng.choice([16, 64, 128]))
    if route == "h":
        B = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)); A = (B + B.conj().T) / 2
        a = np.asfortranarray(np.triu(A)); z = np.zeros((n, n), dtype=np.complex128, order="F")
    else:
        B = rng.standard_normal((n, n)); A = (B + B.T) / 2
        a = np.asfortranarray(np.triu(A)); z = np.zeros((n, n), order="F")
    w = np.zeros(n)
    fn = {"sx": ee.eigen_sx, "s": ee.eigen_s, "h": ee.eigen_h}[route]
    fn(n, nvec, a, n, w, z, n, m_forward=mf, m_backward=mb, mode=mode)
    assert api.last_status() == 0, (it, n, route, mode, api.last_status())
    wr = np.linalg.eigvalsh(A)
    werr = np.abs(w - wr).max() / max(1.0, np.abs(wr).max())
    res = orth = 0.0
    if mode != "N":
        Z = z[:, :nvec]
        res = np.linalg.norm(A @ Z - Z * w[None, :nvec]) / (n * EPS * max(np.linalg.norm(A), 1e-300))
        orth = np.linalg.norm(Z.conj().T @ Z - np.eye(nvec)) / (n * EPS)
    worst = [max(worst[0], werr), max(worst[1], res), max(worst[2], orth)]
    assert werr < 1e-12 and res < 768 and orth < 8, (it, n, route, mode, nvec, mf, mb, werr, res, orth)
print(f"stress OK: {ncalls} solves, worst eigenvalue error {worst[0]:.2e}, residual metric {worst[1]:.3f} (<768), "
      f"orthogonality metric {worst[2]:.3f} (<8)")
