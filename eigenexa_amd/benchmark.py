"""eigenexa_benchmark for the MI355X build: runs the reference's benchmark input files against libeigenexa_amd.

    python -m eigenexa_amd.benchmark [-f IN] [-c | -n] [-L]

Same input-file format as the reference driver (benchmark/IN, benchmark/main2.f:262-300): one case per line,
``N nvec bx by mode matrix solver check``; lines starting with ``!`` are comments; a non-positive N ends the run.

    mode   0 'N' eigenvalues only | 1 'A' all eigenpairs | 2 'X' = 'A' + eigenvalue refinement | 3 'S' | 4 'T' | 5 'C'
           (benchmark/main2.f:327-346)
    matrix 0 Frank | 1 Toeplitz | 2 random | 3 Frank 2 | 4..9 prescribed spectra through the Helmert matrix
           (benchmark/mat_set.f:566-595; 10 / -1 / -2 read files and are not supported here)
    solver 0 eigen_sx (pentadiagonal route) | 1 eigen_s (tridiagonal route)
    check  1: eigenvalue test against the known spectrum (benchmark/w_test.f:141-170) and, for full eigenvector
           sets, residual / orthogonality test (benchmark/ev_test.f:181-204)

The matrix stays in HBM (device API of the C-ABI); the report lines follow the reference's (elapsed time, FLOP,
GFLOPS, the PASSED / CAUTION / FAILED verdicts with the same thresholds).  Options -g / -x (process-grid shapes of
the MPI build) do not apply to the one-GPU driver.
"""
import argparse
import sys
import time

import numpy as np

from . import _lib, api, layout

MODES = {0: "N", 1: "A", 2: "X", 3: "S", 4: "T", 5: "C"}
MODE_TEXT = {
    "N": "mode 'N' :: only eigenvalues, no eigenvector",
    "A": "mode 'A' :: all the eigenpairs",
    "X": "mode 'X' :: mode 'A' + accuracy improvement",
    "S": "mode 'S' :: skip DC but set Z as Identity",
    "T": "mode 'T' :: run DC but skip TRBAK",
    "C": "mode 'C' :: skip DC and TRBAK return X=identity",
}
MATRIX_TEXT = {
    0: "(Frank matrix)", 1: "(Toeplitz matrix)", 2: "(Random matrix)", 3: "(Frank matrix 2)",
    4: "(W: 0, 1, ..., n-1)", 5: "(W: sin(PAI*5*i/(n-1)+EPS^1/4)^3)", 6: "(W: MOD(i,5)+MOD(i,2))",
    7: "(W: same as Frank matrix)", 8: "(W: Uniform Distribution, [0,1))", 9: "(W: Gauss Distribution, m=0,s=1)",
}
EPS = np.finfo(np.float64).eps
EPS2 = np.sqrt(EPS)
EPS4 = np.sqrt(EPS2)


def parse_input(path):
    """yields (n, nvec, bx, by, mode, matrix, solver, check) tuples; stops at n <= 0"""
    with open(path) as f:
        for line in f:
            if not line.strip() or line.lstrip().startswith("!"):
                continue
            v = [int(x) for x in line.split()[:8]]
            if len(v) < 8:
                v += [0] * (8 - len(v))
            if v[0] <= 0:
                return
            yield tuple(v)


def _verdict(x):
    return "PASSED" if x < EPS2 else ("CAUTION" if x < EPS4 else "FAILED")


def w_test(w, lam, out):
    """benchmark/w_test.f:103-170: relative and absolute eigenvalue error against the known spectrum"""
    if lam is None:
        out("*** Eigenvalue Error Test *** : SKIP (no analytic spectrum for this matrix type)")
        return True
    lam = np.sort(lam)
    y = np.abs(w - lam)
    nz = lam != 0.0
    ax = float((y[nz] / np.abs(lam[nz])).max()) if nz.any() else 0.0
    bx = float(y.max())
    amin, amax = float(np.abs(lam).min()), float(np.abs(lam).max())
    out(f"cond(A)=|w_max|/|w_min|= {amax:.6e} / {amin:.6e}")
    out(f"max|w(i)-w(i).true|/|w.true|= {ax:.6e}")
    out(f"*** Eigenvalue Relative Error *** : {_verdict(ax)}")
    out(f"max|w(i)-w(i).true|         = {bx:.6e}")
    out(f"*** Eigenvalue Absolute Error *** : {_verdict(bx)}")
    if bx >= EPS4 and ax < EPS4:
        out(" Do not mind it. Relative error is small enough.")
    return ax < EPS4 or bx < EPS4


def ev_test(A_dev, w_dev, z_dev, n, out):
    """benchmark/ev_test.f:181-204: ||AZ - ZW||_F / (N eps ||A||_F) < 768 ... and ||Z^T Z - I||_F / (N eps) < 8"""
    import torch

    Z = z_dev
    anorm = torch.linalg.norm(A_dev).item()
    res = torch.linalg.norm(A_dev @ Z - Z * w_dev[None, :]).item()
    r = res / (n * EPS * anorm) if anorm > 0 else 0.0
    o = torch.linalg.norm(Z.T @ Z - torch.eye(n, dtype=torch.float64, device=Z.device)).item() / (n * EPS)
    out(f"|A|_{{F}}= {anorm:.6e}")
    out(f"|AZ-ZW|_{{F}}/(N*eps*|A|_{{F}})= {r:.6e}")
    out(f"*** Residual Error Test ***   : {'PASSED' if r < 768 else 'FAILED'}")
    out(f"|ZZ-I|_{{F}}/(N*eps)= {o:.6e}")
    out(f"*** Orthogonality  Test ***   : {'PASSED' if o < 8 else 'FAILED'}")
    return r < 768 and o < 8


def run_case(case, check_default=None, out=print):
    """one input line; returns a dict with the timings and verdicts"""
    import torch

    n, nvec, bx, by, imode, mtype, solver, merror = case
    nvec = min(nvec, n)
    mode = MODES.get(imode, "A")
    check = (merror == 1) if check_default is None else check_default
    if mtype not in MATRIX_TEXT:
        raise ValueError(f"matrix type {mtype} is not supported by this driver")
    dev = torch.device("cuda:0")
    A, lam = layout.reference_matrix(n, mtype)
    nx, ny = api.eigen_get_matdims(n)
    A_dev = torch.from_numpy(A).to(dev)
    a = torch.zeros(ny, nx, dtype=torch.float64, device=dev)   # column-major (nx, ny)
    a[:n, :n] = A_dev.T
    z = torch.zeros(ny, nx, dtype=torch.float64, device=dev)
    w = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    (api.eigen_sx if solver == 0 else api.eigen_s)(n, nvec, a, nx, w, z, nx, m_forward=bx, m_backward=by, mode=mode)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    status = api.last_status()
    flops = float(a[0, 0].item()) if n >= 1 else 0.0
    out("======================================================")
    out("Solver = eigen_sx / via penta-diagonal format" if solver == 0 else "Solver = eigen_s  / via tri-diagonal format")
    out(f"Block width = {bx} / {by}")
    out("NUM.OF.PROCESS= 1 ( 1 1 )   [1x MI355X]")
    out(f"Matrix dimension = {n}")
    out(f"Matrix type = {mtype} {MATRIX_TEXT[mtype]}")
    out(f"The number of eigenvectors computed = {nvec}")
    out(MODE_TEXT[mode])
    out(f"Elapsed time = {elapsed:.6f} [sec]")
    out(f"FLOP         = {abs(flops):.6e}")
    out(f"Performance  = {abs(flops) / elapsed * 1e-9:.3f} [GFLOPS]")
    res = {"n": n, "mode": mode, "solver": solver, "mtype": mtype, "elapsed": elapsed, "status": status, "ok": status == 0}
    if check and status == 0:
        wh = w.cpu().numpy()
        res["w_ok"] = w_test(wh, lam, out)
        res["ok"] = res["ok"] and res["w_ok"]
        if mode in ("A", "X") and nvec == n:
            res["ev_ok"] = ev_test(A_dev, w, z[:n, :n].T, n, out)
            res["ok"] = res["ok"] and res["ev_ok"]
    out("======================================================")
    out("")
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(prog="eigenexa_benchmark", description=__doc__.split("\n\n")[0])
    ap.add_argument("-f", dest="input_file", default="IN", help="input file (default ./IN)")
    ap.add_argument("-c", dest="check", action="store_true", default=None, help="check accuracy for every case")
    ap.add_argument("-n", dest="nocheck", action="store_true", help="never check accuracy")
    ap.add_argument("-L", dest="list", action="store_true", help="list the test matrices and exit")
    args = ap.parse_args(argv)
    if args.list:
        for k in sorted(MATRIX_TEXT):
            print(f" Matrix type = {k:3d} {MATRIX_TEXT[k]}")
        return 0
    check_default = True if args.check else (False if args.nocheck else None)
    lib = _lib.load()
    api.eigen_init()
    ver = np.zeros(1, dtype=np.int32)
    print(f" INPUT FILE='{args.input_file}'")
    bad = 0
    for case in parse_input(args.input_file):
        r = run_case(case, check_default)
        bad += 0 if r["ok"] else 1
    api.eigen_free()
    print(" Benchmark completed" + (f" ({bad} case(s) did not pass)" if bad else ""))
    del lib, ver
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
