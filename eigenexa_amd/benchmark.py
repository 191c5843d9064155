"""eigenexa_benchmark for the MI355X build: runs the reference's benchmark input files against libeigenexa_amd.

    python -m eigenexa_amd.benchmark [-f IN] [-c | -n] [-L] [-g R|C|A|<k>] [-x Px Py]
    python -m torch.distributed.run --nproc-per-node P -m eigenexa_amd.benchmark ...      (one process per GPU)

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
GFLOPS, the PASSED / CAUTION / FAILED verdicts with the same thresholds).

Multi-rank runs (one process per GPU, launched by torch.distributed.run) take the reference's process-grid options
(benchmark/main2.f:139-216): ``-g R`` / ``-g C`` row- / column-major rank order, ``-g A`` every rank solves alone on
its own 1x1 grid (the MPI_COMM_SELF case), ``-g <k>`` split the ranks into k groups of which group 0 solves and the
others do not participate (the MPI_COMM_NULL case), ``-x Px Py`` an explicit Px x Py grid (Px <= Py).  The local
2-D cyclic blocks are filled from the same matrix generators; the checks gather the eigenvector blocks onto rank 0.
``EIGX_BENCH_BACKEND=gloo`` runs all ranks on GPU 0 over the hipIpc peer-window transport (functional rehearsal only).
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
    10: "(W: Read from the data file 'W.dat')",
    -1: "(Read from the data file 'A.mtx')", -2: "(Read from the data file 'B.mtx')",
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


def run_case(case, check_default=None, out=print, mr=None):
    """one input line; returns a dict with the timings and verdicts.  ``mr`` (multi-rank runs) = dict with the
    torch.distributed module, the solver group, this rank's index in it and its size"""
    import torch

    n, nvec, bx, by, imode, mtype, solver, merror = case
    if mtype < 0:   # the file decides the order (mat_dim_get, benchmark/main2.f:366-374)
        n = layout.matrix_market_dim("A.mtx" if mtype == -1 else "B.mtx")
    nvec = min(nvec, n)
    mode = MODES.get(imode, "A")
    check = (merror == 1) if check_default is None else check_default
    if mtype not in MATRIX_TEXT:
        raise ValueError(f"matrix type {mtype} is not supported by this driver")
    dev = torch.device("cuda", torch.cuda.current_device())
    A, lam = layout.reference_matrix(n, mtype)
    nx, ny = api.eigen_get_matdims(n)
    procs, Px, Py = api.eigen_get_procs()
    _, xi, yi = api.eigen_get_id()
    px, py = xi - 1, yi - 1
    rows = np.arange(px, n, Px)      # global indices of this rank's cyclic block (src/eigen_libs0.F:1825-2258)
    cols = np.arange(py, n, Py)
    a = torch.zeros(ny, nx, dtype=torch.float64, device=dev)   # column-major (nx, ny)
    a[: len(cols), : len(rows)] = torch.from_numpy(np.ascontiguousarray(A[np.ix_(rows, cols)].T)).to(dev)
    z = torch.zeros(ny, nx, dtype=torch.float64, device=dev)
    w = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    if mr is not None:
        mr["dist"].barrier(group=mr["group"])
    t0 = time.perf_counter()
    (api.eigen_sx if solver == 0 else api.eigen_s)(n, nvec, a, nx, w, z, nx, m_forward=bx, m_backward=by, mode=mode)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    status = api.last_status()
    lead = mr is None or mr["rank"] == 0   # rank 0 of the solver group runs the checks
    flops = float(a[0, 0].item()) if (n >= 1 and px == 0 and py == 0) else 0.0
    out("======================================================")
    out("Solver = eigen_sx / via penta-diagonal format" if solver == 0 else "Solver = eigen_s  / via tri-diagonal format")
    out(f"Block width = {bx} / {by}")
    out(f"NUM.OF.PROCESS= {procs} ( {Px} {Py} )   [{procs}x MI355X]")
    out(f"Matrix dimension = {n}")
    out(f"Matrix type = {mtype} {MATRIX_TEXT[mtype]}")
    out(f"The number of eigenvectors computed = {nvec}")
    out(MODE_TEXT[mode])
    out(f"Elapsed time = {elapsed:.6f} [sec]")
    out(f"FLOP         = {abs(flops):.6e}")
    out(f"Performance  = {abs(flops) / elapsed * 1e-9:.3f} [GFLOPS]")
    res = {"n": n, "mode": mode, "solver": solver, "mtype": mtype, "elapsed": elapsed, "status": status, "ok": status == 0}
    if check and status == 0:
        want_ev = mode in ("A", "X") and nvec == n
        Zfull = None
        if want_ev and mr is not None:
            # gather the cyclic eigenvector blocks onto every rank of the solver group (driver sizes only)
            bxm, bym = (n + Px - 1) // Px, (n + Py - 1) // Py
            zl = torch.zeros(bym, bxm, dtype=torch.float64, device=dev)
            zl[: len(cols), : len(rows)] = z[: len(cols), : len(rows)]
            zl = zl.cpu() if mr["host_gather"] else zl
            blocks = [torch.zeros_like(zl) for _ in range(mr["size"])]
            mr["dist"].all_gather(blocks, zl, group=mr["group"])
            ids = torch.tensor([px, py], dtype=torch.int64)
            ids = ids if mr["host_gather"] else ids.to(dev)
            idl = [torch.zeros_like(ids) for _ in range(mr["size"])]
            mr["dist"].all_gather(idl, ids, group=mr["group"])
            if lead:
                Zfull = torch.zeros(n, n, dtype=torch.float64, device=dev)   # Zfull[j, i] = Z(i, j)
                for b, idv in zip(blocks, idl):
                    qx, qy = int(idv[0]), int(idv[1])
                    r_q, c_q = np.arange(qx, n, Px), np.arange(qy, n, Py)
                    Zfull[torch.as_tensor(c_q, device=dev)[:, None], torch.as_tensor(r_q, device=dev)[None, :]] = \
                        b[: len(c_q), : len(r_q)].to(dev)
        elif want_ev:
            Zfull = z[:n, :n]
        if lead:
            wh = w.cpu().numpy()
            res["w_ok"] = w_test(wh, lam, out)
            res["ok"] = res["ok"] and res["w_ok"]
            if want_ev:
                res["ev_ok"] = ev_test(torch.from_numpy(A).to(dev), w, Zfull.T, n, out)
                res["ok"] = res["ok"] and res["ev_ok"]
    out("======================================================")
    out("")
    return res


def main(argv=None):
    import os

    ap = argparse.ArgumentParser(prog="eigenexa_benchmark", description=__doc__.split("\n\n")[0])
    ap.add_argument("-f", dest="input_file", default="IN", help="input file (default ./IN)")
    ap.add_argument("-c", dest="check", action="store_true", default=None, help="check accuracy for every case")
    ap.add_argument("-n", dest="nocheck", action="store_true", help="never check accuracy")
    ap.add_argument("-L", dest="list", action="store_true", help="list the test matrices and exit")
    ap.add_argument("-g", dest="grid", default=None, help="R | C rank order, A every rank alone, <k> k groups (group 0 solves)")
    ap.add_argument("-x", dest="dims", nargs=2, type=int, default=None, metavar=("PX", "PY"), help="explicit process grid")
    args = ap.parse_args(argv)
    if args.list:
        for k in sorted(MATRIX_TEXT):
            print(f" Matrix type = {k:3d} {MATRIX_TEXT[k]}")
        return 0
    check_default = True if args.check else (False if args.nocheck else None)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    mr = None
    participant = True
    if world > 1:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EIGX_BENCH_BACKEND", "nccl")
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
        grid = (args.grid or "C")[:1]
        if args.dims is not None:
            pxd, pyd = args.dims
            if pxd * pyd != world:
                if rank == 0:
                    print("Illegal dimensions are specified.")
                dist.destroy_process_group()
                return 1
            if pxd > pyd:
                if rank == 0:
                    print("This process map is not supported.\nPx should be smaller than Py")
                dist.destroy_process_group()
                return 1
            # MPI_Cart_create numbers the ranks of a cartesian communicator row-major
            api.eigen_init(comm=True, order="R", device=local_rank, dims=(pxd, pyd))
            mr = {"dist": dist, "group": None, "rank": rank, "size": world}
        elif grid in "Aa":
            api.eigen_init(device=local_rank)       # MPI_COMM_SELF: every rank owns a 1x1 grid
        elif grid.isdigit() and int(grid) >= 1:
            k = int(grid)
            members = [r for r in range(world) if r % k == 0]
            group = dist.new_group(ranks=members)   # collective over all ranks
            participant = rank % k == 0             # the others hold MPI_COMM_NULL: every call returns at once
            if participant and len(members) > 1:
                api.eigen_init(comm=group, device=local_rank)
                mr = {"dist": dist, "group": group, "rank": members.index(rank), "size": len(members)}
            elif participant:
                api.eigen_init(device=local_rank)
        else:
            api.eigen_init(comm=True, order="R" if grid in "Rr" else "C", device=local_rank)
            mr = {"dist": dist, "group": None, "rank": rank, "size": world}
        if mr is not None:
            mr["host_gather"] = backend != "nccl"
    else:
        _lib.load()
        api.eigen_init()
    # world rank 0 reports (it is rank 0 of every solver group that exists here)
    out = print if rank == 0 else (lambda *a_, **k_: None)
    out(f" INPUT FILE='{args.input_file}'")
    bad = 0
    if participant:
        for case in parse_input(args.input_file):
            r = run_case(case, check_default, out=out, mr=mr)
            bad += 0 if r["ok"] else 1
        api.eigen_free()
    if world > 1:
        import torch
        import torch.distributed as dist

        t = torch.tensor([float(bad)], dtype=torch.float64)
        t = t if os.environ.get("EIGX_BENCH_BACKEND", "nccl") != "nccl" else t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bad = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    out(" Benchmark completed" + (f" ({bad} case(s) did not pass)" if bad else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
