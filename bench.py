#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: GFLOP/s + wall time of the full eigen_sx solve (reference flop model).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full eigen_sx solve (scaling -> pentadiagonal reduction -> band D&C -> back-transform, all
eigenpairs) with the input matrix already resident in HBM when the timed region starts (K pristine copies are made
beforehand because the solver destroys `a`, exactly like the reference).  value = flops credited by the reference's
own model (4/3 N^3 + counted D&C GEMM flops + 2 nvec N^2, src/eigen_sx.F:165,:248,:285-296) / wall time.

  --gpus 1 : BASELINE.json configs[1], N=8192 random symmetric fp64 on one MI355X.  The line also carries an
             `extra` block with configs[2]'s matrix (N=32768) on the same GPU (the faster of two timed solves, both
             listed), so that north_star's "trailing update >= 70 % of the fp64 MFMA roofline at N=32768 on 1 GPU" is
             timed by the driver's run, and `extra_s` / `extra_n65536` with configs[3] / [4] (one timed solve each).
  --gpus N : BASELINE.json configs[2], N=32768 on the Px x Py grid of the N GPUs -- STRONG scaling (the matrix is
             fixed, `value` is the whole-job rate, the driver forms speed-ups from its own 1/2/4/8 runs).  A is
             distributed 2-D cyclically (nothing replicated), see DESIGN.md section 6.  A small sanity solve runs
             first; if any rank fails in it, every rank falls back to independent replicas of the N=8192 solve and
             the line says so ("scaling": "weak").  --weak selects weak scaling (N = size*sqrt(P)) instead.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_MFMA_PEAK_TF = 78.6   # MI355X fp64 matrix peak (v_mfma_f64_16x16x4_f64 = vector rate)
# the reference itself on the survey box (mpiexec -np 8, OMP_NUM_THREADS=1, N=8192 random, eigen_sx)
REFERENCE_PUBLISHED = {"seconds": 30.1, "gflops": 114.0, "cores": 8, "n": 8192,
                       "where": "BASELINE.md: reference 2.13 built with flang + MKL, 8 host cores of the survey container"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", dest="n", type=int, default=0, help="matrix size (default 8192 on one GPU, 32768 on several)")
    ap.add_argument("--route", default="sx", choices=["sx", "s"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="--gpus 1: skip the extra N=32768 / N=65536 solves")
    ap.add_argument("--cpu-n", type=int, default=0, help="size of the cpu_baseline solve (OpenMP oracle on the host cores); "
                    "0 = the GPU line's N (the headline configuration)")
    ap.add_argument("--mf", type=int, default=0, help="m_forward (panel width) of the main line; default 64 on one GPU (reference default 48; N=8192 reduction on one box: 132.8 / 132.0 / 132.6 / 133.7 ms for 48 / 64 / 96 / 128), 128 on several (N=32768: what the per-rank rehearsal was tuned with)")
    ap.add_argument("--extra-mf", type=int, default=256, help="m_forward of the extra N=32768 solve (K = 512 slabs for the trailing update)")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of the distributed solve")
    ap.add_argument("--weak", action="store_true", help="N>1: weak scaling, N = size*sqrt(P) (size defaults to 8192)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process becomes the launcher.  It touches no GPU (torch is
    not even imported yet), starts N copies of this script as child processes with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set (what `python -m torch.distributed.run --nproc-per-node N` would do), relays their
    output (rank 0 prints the JSON line) and exits with the worst exit code."""
    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "EIGX_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    deadline = None
    import signal

    def leave(signum, _frame):     # an interrupted launcher takes its ranks with it (exact PIDs it started)
        for p_ in procs:
            p_.terminate()
        t_end = time.time() + 10.0
        for p_ in procs:
            try:
                p_.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p_.kill()
        sys.exit(128 + signum)

    signal.signal(signal.SIGINT, leave)
    signal.signal(signal.SIGTERM, leave)
    while procs:
        for p_ in list(procs):
            rc = p_.poll()
            if rc is None:
                continue
            procs.remove(p_)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 1)
                if deadline is None:
                    deadline = time.time() + 60.0    # a rank died: the others get a minute to notice and leave
        if deadline is not None and time.time() > deadline:
            for p_ in procs:
                p_.kill()                             # exact PIDs we started
            for p_ in procs:
                p_.wait()
            procs = []
        time.sleep(0.05)
    sys.exit(worst)


ARGS = parse_args() if __name__ == "__main__" else None
if ARGS is not None and ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
    spawn_ranks(ARGS)          # never returns

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    args = ARGS if ARGS is not None else parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.mf <= 0:
        args.mf = 64 if world == 1 else 128
    if args.gpus != world:
        # a line that says n_gpus = world while the caller asked for --gpus N would void a scaling run
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE = {world}: launch with `python bench.py --gpus N` (spawns the ranks "
              f"itself) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`", file=sys.stderr, flush=True)
        sys.exit(2)
    if os.environ.get("EIGX_BENCH_DRYRUN"):   # launcher plumbing test (tests/test_host.py): no GPU is touched
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "master_port": os.environ.get("MASTER_PORT")}), flush=True)
        sys.exit(0 if os.environ.get("EIGX_BENCH_DRYRUN") != f"fail{rank}" else 3)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # EIGX_BENCH_BACKEND=gloo: functional rehearsal of the N>1 code path with all ranks on GPU 0 (the ranks
        # then talk through hipIpc peer windows on the one card); never used for reported numbers
        backend = os.environ.get("EIGX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import eigenexa_amd as ee
    from eigenexa_amd import _lib, api, layout

    lib = _lib.load()
    for kv in filter(None, os.environ.get("EIGX_TUNE", "").split(",")):   # lab hook: "key=value,..." -> eigx_tune (A/B runs)
        k_, v_ = kv.split("=")
        lib.eigx_tune(int(k_), int(v_))
    replicas = args.replicas or world == 1
    # one GPU: the P = 1 point of the strong-scaling series (the multi-GPU lines solve configs[2]'s fixed N = 32768
    # matrix; its one-GPU time is this line's extra.seconds); independent replicas are weak scaling by construction
    scaling = "strong" if world == 1 else "weak"
    if world == 1 or args.replicas:
        n = args.n or 8192
    elif args.weak:
        n = int(round((args.n or 8192) * (world ** 0.5) / 128.0)) * 128
    else:
        n = args.n or 32768
        scaling = "strong"

    def gen_local(nn, Px, Py, px, py):
        # this rank's 2-D cyclic block of the global matrix, generated on the GPU (bit-identical to the numpy
        # generator layout.random_symmetric, see tests/test_host.py)
        rows = np.arange(px, nn, Px)
        cols = np.arange(py, nn, Py)
        return layout.random_symmetric_torch(nn, dev, rows=rows, cols=cols), rows, cols

    def all_ok(local_ok):
        """every rank takes the same branch: a rank-local failure of any kind is folded into one flag and agreed on"""
        if dist is None:
            return bool(local_ok)
        flag = torch.tensor([1.0 if local_ok else 0.0], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return flag.item() == 1.0

    mg_note = ""
    Px = Py = 1
    px = py = 0
    if not replicas:
        why = ""
        ok = True
        try:
            ee.eigen_init(comm=True, device=dev.index)   # session id broadcast over the process group; peer windows + RCCL
        except Exception as exc_init:
            ok, why = False, f"eigen_init: {exc_init}"
        if all_ok(ok):
            # sanity solve through the distributed path before anything is timed; any rank-local exception
            # (status != 0, OOM, LAPACK) is agreed on collectively so that no rank is left waiting in a collective
            werr = float("nan")
            try:
                procs, Px, Py = ee.eigen_get_procs()
                _, xi, yi = ee.eigen_get_id()
                px, py = xi - 1, yi - 1
                ns = 1024
                loc, rows, cols = gen_local(ns, Px, Py, px, py)
                nxs, nys = ee.eigen_get_matdims(ns)
                a_s = torch.zeros(nys, nxs, dtype=torch.float64, device=dev)
                a_s[: len(cols), : len(rows)] = loc.T
                z_s = torch.zeros(nys, nxs, dtype=torch.float64, device=dev)
                w_s = torch.zeros(ns, dtype=torch.float64, device=dev)
                ee.eigen_sx(ns, ns, a_s, nxs, w_s, z_s, nxs, m_forward=128, m_backward=128)
                if api.last_status() != 0:
                    raise RuntimeError(f"sanity solve status {api.last_status()}")
                wref = np.linalg.eigvalsh(layout.random_symmetric(ns))
                werr = float(np.abs(w_s.cpu().numpy() - wref).max() / np.abs(wref).max())
                if not werr < 1e-12:
                    raise RuntimeError(f"sanity solve inaccurate (werr {werr:.2e})")
            except Exception as exc_s:
                ok, why = False, f"sanity solve: {exc_s}"
            ok = all_ok(ok)
            mg_note = f"distributed sanity solve N=1024: max eigenvalue error {werr:.1e}"
        else:
            ok = False
        if not ok:
            print(f"[bench] rank {rank}: distributed path unavailable ({why or 'another rank failed'}); "
                  f"every rank falls back to independent replicas", file=sys.stderr, flush=True)
            replicas = True
            scaling = "weak"
            n = 8192
            Px = Py = 1
            px = py = 0
            try:
                ee.eigen_free()
            except Exception:
                pass
    if replicas:
        _lib.check(lib.eigx_init(dev.index), "eigx_init")   # every rank owns a 1x1 grid on its own GPU

    fn = lib.eigx_sx_dev if args.route == "sx" else lib.eigx_s_dev

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def read_prof():
        prof = np.zeros(6)
        lib.eigx_profile_read(prof.ctypes.data_as(C.POINTER(C.c_double)))
        return prof

    def pmc_traffic(nn, alg_per_launch):
        try:
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            src = next(f for f in ("r04_symv_traffic.json", "r03_symv_traffic.json") if os.path.exists(os.path.join(pdir, f)))
            with open(os.path.join(pdir, src)) as fh:
                pm = json.load(fh)
            if int(pm["n"]) != int(nn):
                return {"traffic": None}
            ratio = float(pm["traffic_over_algorithmic"])
            # nothing of this is measured in THIS run: `traffic` stays null, the committed PMC passes are quoted beside it
            return {"traffic": None,
                    "traffic_from_profile": round(ratio * alg_per_launch, 1),
                    "traffic_source": f"profiles/{src}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                      f"one N={nn} reduction of that round's kernel (gfx950 correction of the guide applied), {ratio:.3f} x "
                                      f"the algorithmic bytes; not collected in this run"}
        except Exception:
            return {"traffic": None}

    def roofline_blocks(prof, nn):
        out = {}
        if prof[0] > 0 and prof[2] > 0:
            ach = prof[1] / prof[2] / 1e9
            out["roofline"] = {
                "kernel": "symv_kernel (fused upper-triangle symmetric mat-vec, 2 vectors)",
                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4),
                # HBM-side bytes need PMC passes of their own (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot run inside
                # this process): taken from the committed passes of the same kernel at the same N when there are any
                # (measured ratio to the algorithmic bytes x this run's algorithmic bytes per launch), else null
                **pmc_traffic(nn, prof[1] / prof[0]),
                "launches_sampled": int(prof[0]), "avg_launch_us": round(prof[2] / prof[0] * 1e6, 2),
                "algorithmic_bytes_per_launch": round(prof[1] / prof[0], 1), "n": nn,
            }
        if prof[3] > 0 and prof[5] > 0:
            ach = prof[4] / prof[5] / 1e12
            out["roofline_trailing_update"] = {
                "kernel": "gemm2_kernel<N,T> tri (rank-2k trailing update, LDS-DMA ring)",
                "bound": "mfma", "achieved": round(ach, 2), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(ach / FP64_MFMA_PEAK_TF, 4), "traffic": None,
                "launches": int(prof[3]), "avg_launch_us": round(prof[5] / prof[3] * 1e6, 2), "n": nn,
            }
        return out

    nx, ny = ee.eigen_get_matdims(n)
    loc, rows, cols = gen_local(n, Px, Py, px, py)
    A_loc_T = loc.T.contiguous()   # [local col, local row]
    del loc
    nrun = args.warmup + args.steps
    a_bufs = []
    for _ in range(nrun):
        a = torch.zeros(ny, nx, dtype=torch.float64, device=dev)
        a[: len(cols), : len(rows)] = A_loc_T
        a_bufs.append(a)
    z = torch.zeros(ny, nx, dtype=torch.float64, device=dev)
    w = torch.zeros(n, dtype=torch.float64, device=dev)

    def solve(a):
        rc = fn(n, n, a.data_ptr(), nx, w.data_ptr(), z.data_ptr(), nx, args.mf, 128, b"A")
        _lib.check(rc, "eigen_" + args.route)
        if os.environ.get("EIGX_BENCH_VERBOSE"):   # lab: stage timers of every solve (warm-up included) on stderr
            tv = np.zeros(16)
            lib.eigx_get_timers(tv.ctypes.data_as(C.POINTER(C.c_double)))
            print(f"[bench] solve: total {tv[0]*1e3:.1f} ms, reduction {tv[1]*1e3:.1f}, dc {tv[2]*1e3:.1f}, bt {tv[3]*1e3:.1f}",
                  file=sys.stderr, flush=True)

    torch.cuda.synchronize()
    for i in range(args.warmup):
        solve(a_bufs[i])
    # bracket every 32nd SYMV launch and every trailing-update launch with HIP events: the events cost time themselves
    # (N = 8192 reduction 126.4 ms without, 130.2 ms with every 32nd launch bracketed), so the sample is kept sparse
    lib.eigx_profile(32)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        solve(a_bufs[args.warmup + i])
    barrier()
    dt = time.perf_counter() - t0
    prof = read_prof()
    kinds = np.zeros(18)
    lib.eigx_profile_read_kinds(kinds.ctypes.data_as(C.POINTER(C.c_double)), 6)
    lib.eigx_profile(0)
    tm = np.zeros(16)
    lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
    flops_one = abs(float(tm[12]))  # = a(1,1): flops credited by the reference model

    if dist is not None:
        cpu_coll = dist.get_backend() != "nccl"
        t = torch.tensor([dt, float(tm[4])], dtype=torch.float64, device="cpu" if cpu_coll else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        comm_max = float(t[1].item())
    else:
        comm_max = 0.0

    # accuracy of the last solve (outside the timed region): the reference's gates
    eps = np.finfo(np.float64).eps
    acc = {}
    if replicas:
        Z = z[:n, :n].T
        Afull = A_loc_T.T
        anorm = torch.linalg.norm(Afull).item()
        res_abs = torch.linalg.norm(Afull @ Z - Z * w[None, :]).item()
        acc = {"residual_over_anorm": res_abs / anorm,
               "residual_metric_lt_768": round(res_abs / (n * eps * anorm), 5),
               "orthogonality_metric_lt_8": round(
                   torch.linalg.norm(Z.T @ Z - torch.eye(n, dtype=torch.float64, device=dev)).item() / (n * eps), 5)}
        del Z, Afull
    else:
        # distributed checks on the cyclic blocks: trace and Frobenius norm of A against the eigenvalues, and the
        # residual ||A Z - Z W||_F through a 2-D SUMMA-free formulation: every rank multiplies ITS block of A with
        # the matching rows of Z gathered along its grid row ... kept simple: rank-local pieces of the invariants
        cpu_coll = dist.get_backend() != "nccl"
        tr = torch.zeros(2, dtype=torch.float64, device=dev)
        rr = torch.from_numpy(rows).to(dev)
        cc = torch.from_numpy(cols).to(dev)
        Aloc = A_loc_T.T  # [local row, local col]
        diag_mask = rr[:, None] == cc[None, :]
        tr[0] = (Aloc * diag_mask).sum()
        tr[1] = (Aloc * Aloc).sum()
        # column norms of my block of Z: sum over the grid column gives ||z_k||^2 = 1 for my columns
        zl = z[: len(cols), : len(rows)]
        zn = (zl * zl).sum(dim=1)
        trc = tr.cpu() if cpu_coll else tr
        dist.all_reduce(trc)
        anorm = float(trc[1].sqrt().item())
        znorm_full = torch.zeros(n, dtype=torch.float64, device=dev)
        znorm_full[cc] = zn
        zc = znorm_full.cpu() if cpu_coll else znorm_full
        dist.all_reduce(zc)
        # randomised residual: || A X - Z W Z^T X ||_F / (||A||_F ||X||_F) for 4 random vectors X -- three block mat-vecs
        # and three allreduces of n x 4 numbers on the 2-D cyclic blocks; a wrong or non-orthogonal Z shows up here
        def allsum(t):
            tc = t.cpu() if cpu_coll else t
            dist.all_reduce(tc)
            return tc.to(dev)

        gx = torch.Generator(device="cpu").manual_seed(7)
        X = torch.randn(n, 4, dtype=torch.float64, generator=gx).to(dev)
        Zb = zl.T                                            # [local row, local col]
        t1 = torch.zeros(n, 4, dtype=torch.float64, device=dev)
        t1[cc] = Zb.T @ X[rr]
        t1 = allsum(t1) * w[:, None]
        y2 = torch.zeros(n, 4, dtype=torch.float64, device=dev)
        y2[rr] = Zb @ t1[cc]
        y2 = allsum(y2)
        y1 = torch.zeros(n, 4, dtype=torch.float64, device=dev)
        y1[rr] = Aloc @ X[cc]
        y1 = allsum(y1)
        probe = float(torch.linalg.norm(y1 - y2).item()) / (anorm * float(torch.linalg.norm(X).item()))
        acc = {"probe_residual_over_anorm": probe,
               "trace_error_over_anorm": abs(float(w.sum().item()) - float(trc[0].item())) / anorm,
               "frobenius_error_over_anorm": abs(float(torch.linalg.norm(w).item()) - anorm) / anorm,
               "max_abs_znorm2_minus_1": float((zc - 1.0).abs().max().item()),
               "comm_seconds_per_solve_max_over_ranks": round(comm_max, 4),
               "sanity": mg_note,
               # what carried the per-step exchange and the bulk collectives, and the init-time self-test of both transports
               "transport": ee.eigen_comm_info(),
               # rank 0's sampled reduction steps (every 32nd), HIP events on the compute stream, averages in microseconds.
               # One rank per GPU: ONE launch per step ("symv" is then the whole step launch: kl of the previous step | ka over
               # the rank's own rows | local mat-vec; the other entries are None).  Ranks sharing a card (wait kernels) or the
               # collective exchanges: role by role -- local mat-vec | sums of the tile partial sums to the row owners |
               # wait for the Y messages | ka over the rank's own rows | wait for the X messages
               "per_step_us": {name: (round(kinds[3 * k_ + 2] / kinds[3 * k_] * 1e6, 2) if kinds[3 * k_] > 0 else None)
                               for k_, name in ((0, "symv"), (2, "exchange_reduce_and_push"), (3, "wait"), (4, "ka"), (5, "wait_x"))},
               "per_step_samples": int(kinds[0])}
    total_flops_all = flops_one * args.steps * (world if replicas else 1)

    out = None
    if rank == 0:
        value = total_flops_all / dt / 1e9
        if world == 1:
            par = "1 GPU (1x1 grid)"
        elif replicas:
            par = f"{world} independent replicas of the N={n} solve (fallback: the distributed path was not usable)"
        else:
            par = (f"{world} GPUs, {Px}x{Py} grid, A 2-D cyclic and sharded (N^2*8/P bytes per GPU, used in place); per step two "
                   f"exchanges: the locally reduced mat-vec sums go to the rank that owns the row for the panel work, the owners "
                   f"send the new x and W rows to everybody (config.transport says how: kernel stores over xGMI into "
                   f"hipIpc-mapped windows, or allgathers), panel gather on a side stream under "
                   f"the local trailing update (look-ahead), D&C row-distributed, back-transformation column-parallel with "
                   f"streamed reflector panels; {scaling} scaling; the one-GPU time of this matrix is extra.seconds of the "
                   f"--gpus 1 line")
        out = {
            "metric": "eigen_sx full-solve throughput (reference flop model: 4/3 N^3 + D&C GEMM + 2 nvec N^2)"
            if args.route == "sx" else "eigen_s full-solve throughput (reference flop model)",
            "value": round(value, 1),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"N={n} random symmetric fp64 (counter-based R+R^T, seed 20240807), eigen_{args.route} "
                            f"all eigenpairs, m_forward={args.mf}, m_backward=128",
                "parallelism": par,
                "stage_ms": {"reduction": round(tm[1] * 1e3, 2), "dc": round(tm[2] * 1e3, 2),
                             "backtransform": round(tm[3] * 1e3, 2)},
                **acc,
            },
        }
        out.update(roofline_blocks(prof, n))

    # ---- extra blocks (one GPU): driver-timed solves of the other BASELINE configs on this GPU -----------------
    #   extra          configs[2]'s matrix, N=32768 eigen_sx all eigenpairs (north_star's 70 % trailing-update target)
    #   extra_s        configs[3]'s, N=32768 eigen_s (tridiagonal route) all eigenpairs
    #   extra_n65536   configs[4]'s, N=65536 eigenvalues only (mode 'N': reduction + bisection)
    def gen_big(n2):
        # leading dimension as a caller of the reference API sizes it: eigen_get_matdims (channel-friendly since round 3,
        # DESIGN.md section 2); beyond the reference's 32-bit guard (N = 65536) the same rule by hand
        lda2 = ee.eigen_get_matdims(n2)[0]
        if lda2 < n2:
            lda2 = (n2 + 95) // 32 * 32
            if lda2 % 2048 < 512:
                lda2 += 512 - lda2 % 2048
            elif lda2 % 2048 > 1536:
                lda2 += 2048 - lda2 % 2048 + 512
        a2 = torch.empty(n2, lda2, dtype=torch.float64, device=dev)
        a2[:, n2:] = 0.0
        fro2 = 0.0
        tr2 = 0.0
        for c0 in range(0, n2, 4096):
            blk = layout.random_symmetric_torch(n2, dev, rows=np.arange(n2), cols=np.arange(c0, c0 + 4096))
            a2[c0:c0 + 4096, :n2] = blk.T
            fro2 += float((blk * blk).sum().item())
            tr2 += float(torch.diagonal(blk[c0:c0 + 4096, :]).sum().item())
            del blk
        return a2, lda2, fro2, tr2

    def extra_block(n2, route, mode, mf2, warm, timed=1):
        fn2 = lib.eigx_sx_dev if route == "sx" else lib.eigx_s_dev
        a2, lda2, fro2, tr2 = gen_big(n2)
        nvec2 = n2 if mode == "A" else 0
        w2 = torch.zeros(n2, dtype=torch.float64, device=dev)
        z2 = torch.empty(n2, lda2, dtype=torch.float64, device=dev) if nvec2 else torch.zeros(8, dtype=torch.float64, device=dev)
        if warm:
            a2w = a2.clone()                                   # pristine copy for the warm-up (the solver destroys a)
            _lib.check(fn2(n2, nvec2, a2w.data_ptr(), lda2, w2.data_ptr(), z2.data_ptr(), lda2, mf2, 128, mode.encode()),
                       "eigen_" + route)                       # warm-up: workspace allocation (~26 GB) happens here
            del a2w
        # `timed` solves, each with its own events; the block reports the median (of two: the slower) and lists all (the first ~40 s of work on
        # a freshly acquired card run on the part's lower bandwidth level)
        runs = []
        for t_ in range(timed):
            a_run = a2 if t_ == timed - 1 else a2.clone()      # the solver destroys its input
            lib.eigx_profile(8)
            torch.cuda.synchronize()
            t0x = time.perf_counter()
            _lib.check(fn2(n2, nvec2, a_run.data_ptr(), lda2, w2.data_ptr(), z2.data_ptr(), lda2, mf2, 128, mode.encode()),
                       "eigen_" + route)
            torch.cuda.synchronize()
            dtx = time.perf_counter() - t0x
            profx = read_prof()
            lib.eigx_profile(0)
            tmx = np.zeros(16)
            lib.eigx_get_timers(tmx.ctypes.data_as(C.POINTER(C.c_double)))
            runs.append((dtx, profx, tmx))
            if a_run is not a2:
                del a_run
        # the block reports the MEDIAN solve (round 4; rounds 1-3 reported the fastest) and lists all of them
        dt2, prof2, tm2 = sorted(runs, key=lambda r_: r_[0])[(len(runs) - 1) // 2 if len(runs) % 2 else len(runs) // 2]
        anorm2 = fro2 ** 0.5
        what = "all eigenpairs" if mode == "A" else "eigenvalues only (mode 'N': reduction + multi-section bisection)"
        ex = {"workload": f"N={n2} random symmetric fp64, eigen_{route} {what}, m_forward={mf2}, "
                          + ("ONE timed solve" if timed == 1 else f"the {'slower' if timed == 2 else 'median'} of {timed} timed solves (all in seconds_each)")
                          + " on this GPU "
                          + ("after one warm-up solve" if warm else "without a warm-up solve (workspace allocation included)"),
              "seconds": round(dt2, 3), "seconds_each": [round(r_[0], 3) for r_ in runs],
              "gflops": round(abs(float(tm2[12])) / dt2 / 1e9, 1),
              "stage_ms": {"reduction": round(tm2[1] * 1e3, 1), "dc_or_bisection": round(tm2[2] * 1e3, 1),
                           "backtransform": round(tm2[3] * 1e3, 1)},
              "frobenius_error_over_anorm": abs(float(torch.linalg.norm(w2).item()) - anorm2) / anorm2,
              "trace_error_over_anorm": abs(float(w2.sum().item()) - tr2) / anorm2,
              "sorted": bool((w2[1:] >= w2[:-1]).all().item())}
        ex.update(roofline_blocks(prof2, n2))
        del a2, z2, w2
        torch.cuda.empty_cache()
        return ex

    if world == 1 and not args.no_extra and n != 32768 and args.route == "sx":
        del a_bufs, z, w, A_loc_T
        torch.cuda.empty_cache()
        for key, (n2, route2, mode2, mf2, warm2, timed2) in {"extra": (32768, "sx", "A", args.extra_mf, True, 2),
                                                             "extra_s": (32768, "s", "A", min(args.extra_mf, 128), True, 1),   # tridiagonal route: 128 beats 256 by 1.2 %
                                                             "extra_n65536": (65536, "sx", "N", args.extra_mf, False, 1)}.items():
            try:
                out[key] = extra_block(n2, route2, mode2, mf2, warm2, timed2)
            except Exception as exc_x:   # an extra block never invalidates the main line
                out[key] = {"error": str(exc_x)}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            # cpu_baseline: the OpenMP oracle on the headline configuration itself (N = the GPU line's N), after all GPU work,
            # alone on the host cores (beside the GPU's extra blocks its busy-waiting OpenMP barriers and the HIP runtime's
            # threads starved each other under the box's CPU quota: the run was killed after 7 silent minutes)
            from oracle import orc

            nc = args.cpu_n or n
            cores = orc.set_threads(int(os.environ.get("EIGX_CPU_CORES", 0)) or min(orc.host_cores(), 16 * max(1, torch.cuda.device_count())))
            print(f"[bench] GPU part done; cpu_baseline: oracle eigen_{args.route} at N={nc} on {cores} host cores "
                  f"(~100 s at N=8192 on 16 cores) ...", file=sys.stderr, flush=True)
            Ac = layout.random_symmetric(nc)
            t0c = time.perf_counter()
            _, _, stats, st = orc.eigen(Ac, args.route)
            tc = time.perf_counter() - t0c
            out["cpu_baseline"] = {
                "value": round(abs(stats[0]) / tc / 1e9, 3), "unit": "GFLOP/s", "cores": cores,
                "kind": "port", "n": nc,
                "sample": f"oracle/eigx_oracle.c eigen_{args.route} (unblocked C restatement, its O(N^3) loops threaded with "
                          f"OpenMP over {cores} host cores of this box; the only CPU code that can run on the GPU box) on the "
                          f"GPU line's own configuration" + ("" if nc == n else f" scaled down to N={nc} (--cpu-n)") +
                          f": N={nc}, the same generator, all eigenpairs, ONE solve, {tc:.1f} s (reduction {st[0]:.1f} s, "
                          f"D&C {st[1]:.1f} s, back-transform {st[2]:.1f} s); the reference's own MPI CPU path (another box) "
                          f"is in reference_published",
                "reference_published": REFERENCE_PUBLISHED,
            }
        print(json.dumps(out), flush=True)
    lib.eigx_free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
