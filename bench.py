#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: GFLOP/s + wall time of the full eigen_sx solve (reference flop model).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full eigen_sx solve (scaling -> pentadiagonal reduction -> band D&C -> back-transform, all
eigenpairs) of the N=8192 random symmetric fp64 matrix of BASELINE.json configs[1], with the input matrix
already resident in HBM when the timed region starts (K pristine copies are made beforehand because the
solver destroys `a`, exactly like the reference).  value = flops credited by the reference's own model
(4/3 N^3 + counted D&C GEMM flops + 2 nvec N^2, src/eigen_sx.F:165,:248,:285-296) / wall time.

N > 1 in round 1: the 2-D cyclic multi-GPU path (DESIGN.md section e) is not built yet, so every rank solves
an independent replica ("replicas", scaling "weak"); value is the aggregate over replicas and the JSON says so.
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
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--route", default="sx", choices=["sx", "s"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=2048)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import eigenexa_amd as ee
    from eigenexa_amd import _lib, layout

    lib = _lib.load()
    # replicas: every rank owns a 1x1 grid on its own GPU
    _lib.check(lib.eigx_init(dev.index), "eigx_init")

    n = args.n
    nx, ny = ee.eigen_get_matdims(n)
    A_host = layout.random_symmetric(n)
    A_dev = torch.from_numpy(np.ascontiguousarray(A_host.T)).to(dev)  # A_dev[j, i] = A(i, j)
    nrun = args.warmup + args.steps
    a_bufs = []
    for _ in range(nrun):
        a = torch.zeros(ny, nx, dtype=torch.float64, device=dev)
        a[:n, :n] = A_dev
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
    flops_one = abs(float(a_bufs[-1][0, 0].item()))  # a(1,1) = flops credited by the reference model

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # accuracy of the last solve (outside the timed region): the reference's gates
    Z = z[:n, :n].T
    eps = np.finfo(np.float64).eps
    Afull = A_dev.T
    anorm = torch.linalg.norm(Afull).item()
    res_abs = torch.linalg.norm(Afull @ Z - Z * w[None, :]).item()
    res_metric = res_abs / (n * eps * anorm)
    orth_metric = torch.linalg.norm(Z.T @ Z - torch.eye(n, dtype=torch.float64, device=dev)).item() / (n * eps)

    if rank == 0:
        total_flops = flops_one * args.steps * world
        value = total_flops / dt / 1e9
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
                "parallelism": "1 GPU (1x1 grid)" if world == 1 else
                               f"{world} independent replicas (2-D cyclic multi-GPU path not built in round 1)",
                "stage_ms": {"reduction": round(tm[1] * 1e3, 2), "dc": round(tm[2] * 1e3, 2),
                             "backtransform": round(tm[3] * 1e3, 2)},
                "residual_over_anorm": res_abs / anorm,
                "residual_metric_lt_768": round(res_metric, 5),
                "orthogonality_metric_lt_8": round(orth_metric, 5),
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
        if prof[3] > 0 and prof[5] > 0:
            ach = prof[4] / prof[5] / 1e12
            out["roofline_trailing_update"] = {
                "kernel": "gemm_f64_kernel<N,T> tri (rank-2k trailing update)",
                "bound": "mfma", "achieved": round(ach, 2), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(ach / FP64_MFMA_PEAK_TF, 4), "traffic": None,
                "launches": int(prof[3]), "avg_launch_us": round(prof[5] / prof[3] * 1e6, 2),
            }
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
