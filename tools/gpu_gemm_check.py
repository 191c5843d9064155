"""GPU check + timing of the fp64 MFMA GEMM through the C-ABI (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigenexa_amd import _lib

lib = _lib.load()
_lib.check(lib.eigx_init(0), "eigx_init")
dev = torch.device("cuda:0")
torch.manual_seed(1)


def colmajor(rows, cols, ld=None):
    ld = ld or rows
    t = torch.randn(cols, ld, dtype=torch.float64, device=dev)  # t[j, i] = element (i, j)
    return t, ld


def run(opa, opb, M, N, K, tri=0, alpha=-1.0, beta=1.0, reps=0):
    Ar, Ac = (M, K) if opa == "N" else (K, M)
    Br, Bc = (K, N) if opb == "N" else (N, K)
    A, lda = colmajor(Ar, Ac, Ar + 3)
    B, ldb = colmajor(Br, Bc, Br + 1)
    Cm, ldc = colmajor(M, N, M + 5)
    C0 = Cm.clone()
    torch.cuda.synchronize()
    b = lambda s: s.encode()
    rc = lib.eigx_dgemm_dev(b(opa), b(opb), M, N, K, alpha, A.data_ptr(), lda, B.data_ptr(), ldb, beta,
                            Cm.data_ptr(), ldc, tri)
    _lib.check(rc, "dgemm")
    Amat = A[:, :Ar].T  # (Ar x Ac)
    Bmat = B[:, :Br].T
    opA = Amat if opa == "N" else Amat.T
    opB = Bmat if opb == "N" else Bmat.T
    ref = alpha * (opA @ opB) + beta * C0[:, :M].T
    got = Cm[:, :M].T
    if tri:
        # tiles strictly below the diagonal are skipped: compare only rows <= cols
        mask = torch.triu(torch.ones(M, N, dtype=torch.bool, device=dev))
        err = ((got - ref) * mask).abs().max().item()
    else:
        err = (got - ref).abs().max().item()
    pad_ok = torch.equal(Cm[:, M:], C0[:, M:])
    scale = ref.abs().max().item()
    msg = f"{opa}{opb} M={M} N={N} K={K} tri={tri}: max err {err:.3e} (scale {scale:.2e}) pad_ok={pad_ok}"
    if reps:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.eigx_dgemm_dev(b(opa), b(opb), M, N, K, alpha, A.data_ptr(), lda, B.data_ptr(), ldb, beta,
                               Cm.data_ptr(), ldc, tri)
        dt = (time.perf_counter() - t0) / reps
        fl = 2.0 * M * N * K * (0.5 if tri else 1.0)
        msg += f"  {dt*1e3:.3f} ms  {fl/dt/1e12:.2f} TFLOP/s"
    print(msg, flush=True)
    assert err < 1e-10 * max(1.0, scale) * max(1, K) ** 0.5 and pad_ok


if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    run("N", "T", 16384, 16384, 256, tri=1, reps=2)
    run("N", "T", 16384, 16384, 1024, tri=0, alpha=-1.0, beta=0.0, reps=2)
    run("N", "N", 8192, 8192, 8192, reps=2)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "exp":
    for kk in (256, 512, 1024):
        run("N", "T", 16384, 16384, kk, tri=0, alpha=-1.0, beta=1.0, reps=4)
        run("N", "T", 16384, 16384, kk, tri=0, alpha=-1.0, beta=0.0, reps=4)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "k1":
    run("N", "T", 1000, 1000, 96, tri=1)
    run("N", "T", 3000, 3000, 64, tri=1)
    for nn in (8192, 16384, 32768):
        for kk in (256, 512):
            run("N", "T", nn, nn, kk, tri=1, reps=4)
    sys.exit(0)
for opa in "NT":
    for opb in "NT":
        run(opa, opb, 300, 200, 77)
        run(opa, opb, 129, 257, 16)
        run(opa, opb, 5, 3, 1)
run("N", "T", 1000, 1000, 96, tri=1)
# performance shapes
run("N", "N", 8192, 8192, 8192, reps=3)
run("T", "N", 512, 8192, 8192, reps=5)   # back-transform W = V^T Z
run("N", "N", 8192, 8192, 512, reps=5)   # back-transform Z -= V W
run("N", "T", 8192, 8192, 256, tri=1, reps=10)   # trailing update, m=128
run("N", "T", 16384, 16384, 256, tri=1, reps=5)
run("N", "T", 16384, 16384, 96, tri=1, reps=5)   # reference default m=48
print("GEMM CHECK PASSED")
