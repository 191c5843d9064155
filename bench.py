#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: GFLOP/s + wall time of the full eigen_sx solve (reference flop model).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full eigen_sx solve (scaling -> pentadiagonal reduction -> band D&C -> back-transform, all
eigenpairs) of the N=8192 random symmetric fp64 matrix of BASELINE.json configs[1], with the input matrix
already resident in HBM when the timed region starts (K pristine copies are made beforehand because the
solver destroys `a`, exactly like the reference).  value = flops credited by the reference's own model
(4/3 N^3 + counted D&C GEMM flops + 2 nvec N^2, src/eigen_sx.F:165,:248,:285-296) / wall time.

N > 1: the sharded multi-GPU path on the 2-D cyclic API layout, weak scaling at fixed memory
per GPU (N = 8192*sqrt(P), rounded to 128); a small sanity solve runs first and, if the multi-GPU path raises, the
run falls back to independent replicas and says so in config.parallelism.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_MFMA_PEAK_TF = 78.6   # MI355X fp64 matrix peak (v_mfma_f64_16x16x4_f64 = vector rate)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", dest="n", type=int, default=8192)
    ap.add_argument("--route", default="sx", choices=["sx", "s"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=2048)
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of the sharded path")
    ap.add_argument("--n-fixed", action="store_true", help="N>1: keep --size (strong scaling) instead of size*sqrt(P)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # EIGX_BENCH_BACKEND=gloo: functional rehearsal of the N>1 code path with all ranks on GPU 0 and the
        # host-staged transport (RCCL refuses duplicate devices); never used for reported numbers
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
    replicas = args.replicas or world == 1
    n = args.n
    if world > 1 and not args.replicas and not args.n_fixed:
        # weak scaling at fixed memory per GPU: N_P = N_1 * sqrt(P), rounded to the 128-column tile
        n = int(round(args.n * (world ** 0.5) / 128.0)) * 128

    def gen_local(nn, Px, Py, px, py):
        # this rank's 2-D cyclic block of the global matrix, generated on the GPU (bit-identical to the numpy
        # generator layout.random_symmetric, see tests/test_host.py)
        rows = np.arange(px, nn, Px)
        cols = np.arange(py, nn, Py)
        return layout.random_symmetric_torch(nn, dev, rows=rows, cols=cols), rows, cols

    mg_note = ""
    if not replicas:
        try:
            init_ok = 1.0
            try:
                ee.eigen_init(comm=True, device=dev.index)   # RCCL world communicator from a broadcast unique id
            except Exception as exc_init:
                print(f"[bench] rank {rank}: eigen_init failed: {exc_init}", file=sys.stderr, flush=True)
                init_ok = 0.0
            # every rank must take the same branch: agree on the outcome before any library collective runs
            flag = torch.tensor([init_ok], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() != 1.0:
                raise RuntimeError("eigen_init failed on at least one rank")
            procs, Px, Py = ee.eigen_get_procs()
            _, xi, yi = ee.eigen_get_id()
            px, py = xi - 1, yi - 1
            # sanity solve through the multi-GPU path before anything is timed
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
            ok = torch.tensor([1.0 if werr < 1e-12 else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() != 1.0:
                raise RuntimeError(f"multi-GPU sanity solve inaccurate (werr {werr:.2e})")
            mg_note = f"multi-GPU sanity N={ns}: max eigenvalue error {werr:.1e}"
        except Exception as exc:  # fall back to replicas, and say so
            print(f"[bench] multi-GPU path unavailable ({exc}); falling back to replicas", file=sys.stderr, flush=True)
            replicas = True
            n = args.n
            try:
                ee.eigen_free()
            except Exception:
                pass
    if replicas:
        _lib.check(lib.eigx_init(dev.index), "eigx_init")   # every rank owns a 1x1 grid on its own GPU
        Px = Py = 1
        px = py = 0

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
    fn = lib.eigx_sx_dev if args.route == "sx" else lib.eigx_s_dev

    def solve(a):
        rc = fn(n, n, a.data_ptr(), nx, w.data_ptr(), z.data_ptr(), nx, 128, 128, b"A")
        _lib.check(rc, "eigen_" + args.route)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    for i in range(args.warmup):
        solve(a_bufs[i])
    lib.eigx_profile(8)  # bracket every 8th SYMV launch and every trailing-update launch with HIP events
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        solve(a_bufs[args.warmup + i])
    barrier()
    dt = time.perf_counter() - t0
    prof = np.zeros(6)
    lib.eigx_profile_read(prof.ctypes.data_as(C.POINTER(C.c_double)))
    lib.eigx_profile(0)
    tm = np.zeros(16)
    lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
    flops_one = abs(float(tm[12]))  # = a(1,1): flops credited by the reference model

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

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
    else:
        # distributed invariants: trace and Frobenius norm of A against the eigenvalues; Z^T Z diagonal sample
        tr = torch.zeros(2, dtype=torch.float64, device=dev)
        rr = torch.from_numpy(rows).to(dev)
        cc = torch.from_numpy(cols).to(dev)
        Aloc = A_loc_T.T  # [local row, local col]
        diag_mask = rr[:, None] == cc[None, :]
        tr[0] = (Aloc * diag_mask).sum()
        tr[1] = (Aloc * Aloc).sum()
        dist.all_reduce(tr)
        anorm = float(tr[1].sqrt().item())
        acc = {"trace_error_over_anorm": abs(float(w.sum().item()) - float(tr[0].item())) / anorm,
               "frobenius_error_over_anorm": abs(float(torch.linalg.norm(w).item()) - anorm) / anorm,
               "sanity": mg_note}
    if replicas:
        total_flops_all = flops_one * args.steps * world
    else:
        total_flops_all = flops_one * args.steps

    if rank == 0:
        value = total_flops_all / dt / 1e9
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"N={n} random symmetric fp64 (counter-based R+R^T, seed 20240807), eigen_{args.route} "
                            f"all eigenpairs, m_forward=128, m_backward=128",
                "parallelism": "1 GPU (1x1 grid)" if world == 1 else (
                    f"{world} independent replicas (fallback)" if replicas else
                    f"{world} GPUs, {Px}x{Py} 2-D cyclic API layout; reduction sharded by 128-column tile ownership "
                    f"(1 RCCL allreduce/step; look-ahead panel bcast on a side stream under the trailing update), D&C GEMMs row-distributed (z allreduce per merge), "
                    f"back-transform column-parallel; "
                    f"weak scaling N = {args.n}*sqrt(P)"),
                "stage_ms": {"reduction": round(tm[1] * 1e3, 2), "dc": round(tm[2] * 1e3, 2),
                             "backtransform": round(tm[3] * 1e3, 2)},
                **acc,
            },
        }
        if prof[0] > 0 and prof[2] > 0:
            ach = prof[1] / prof[2] / 1e9
            out["roofline"] = {
                "kernel": "symv_kernel (fused upper-triangle symmetric mat-vec, 2 vectors)",
                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                "launches_sampled": int(prof[0]), "avg_launch_us": round(prof[2] / prof[0] * 1e6, 2),
                "algorithmic_bytes_per_launch": round(prof[1] / prof[0], 1),
            }
            # HBM-side traffic of the same kernel from the committed PMC passes (profiles/r01_symv_traffic.json:
            # separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
            # 16-byte-per-lane reads on gfx950), expressed like `achieved`: PMC bytes per launch / launch duration
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_symv_traffic.json")))
                ratio = float(tj["traffic_over_algorithmic"])
                out["roofline"]["traffic"] = round(ach * ratio, 1)
                out["roofline"]["traffic_over_algorithmic"] = round(ratio, 3)
                out["roofline"]["traffic_measured_at_n"] = tj["n"]
            except Exception:
                pass
        if prof[3] > 0 and prof[5] > 0:
            ach = prof[4] / prof[5] / 1e12
            out["roofline_trailing_update"] = {
                "kernel": "gemm2_kernel<N,T> tri (rank-2k trailing update, LDS-DMA ring)",
                "bound": "mfma", "achieved": round(ach, 2), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(ach / FP64_MFMA_PEAK_TF, 4), "traffic": None,
                "launches": int(prof[3]), "avg_launch_us": round(prof[5] / prof[3] * 1e6, 2),
            }
            # HBM-side traffic from the committed PMC passes (profiles/r01_trailing_update_traffic.json), at the
            # nearest measured size: fetch (doubled, see above) + write bytes over the algorithmic bytes
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_trailing_update_traffic.json")))
                row = min((r for r in tj["rows"] if r["K"] == 256), key=lambda r: abs(r["n"] - n))
                out["roofline_trailing_update"]["traffic"] = {
                    "traffic_over_algorithmic": row["traffic_over_algorithmic"],
                    "fetch_over_algorithmic": row["fetch_over_algorithmic"],
                    "write_over_algorithmic": row["write_over_algorithmic"], "measured_at_n": row["n"]}
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            from oracle import orc

            nc = args.cpu_n
            Ac = layout.random_symmetric(nc)
            t0c = time.perf_counter()
            _, _, stats, st = orc.eigen(Ac, args.route)
            tc = time.perf_counter() - t0c
            out["cpu_baseline"] = {
                "value": round(abs(stats[0]) / tc / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
                "sample": f"oracle/eigx_oracle.c eigen_{args.route}, N={nc} same generator, all eigenpairs, "
                          f"{tc:.1f} s on one host core (reduction {st[0]:.1f} s, D&C {st[1]:.1f} s, "
                          f"back-transform {st[2]:.1f} s)",
            }
        print(json.dumps(out), flush=True)
    lib.eigx_free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
