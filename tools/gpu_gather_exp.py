import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigenexa_amd import _lib
lib = _lib.load(); _lib.check(lib.eigx_init(0), "init")
dev = torch.device("cuda:0")
def t(M, N, K, ld, opb, label):
    A = torch.randn(8192, ld, dtype=torch.float64, device=dev)
    B = torch.randn(8192, ld, dtype=torch.float64, device=dev)
    C = torch.zeros(8192, ld, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        lib.eigx_dgemm_gather_dev(b"N", opb, M, N, K, 1.0, A.data_ptr(), ld, B.data_ptr(), ld, 0.0, C.data_ptr(), ld, None, None)
        best = min(best, time.perf_counter() - t0)
    print(f"{label} M={M} N={N} K={K} ld={ld}: {best*1e3:.2f} ms {2.0*M*N*K/best/1e12:.1f} TF", flush=True)
for ld in (8192, 8224, 8195):
    t(8192, 6912, 6912, ld, b"T", "NT")
    t(8192, 6912, 6912, ld, b"N", "NN")
    t(8192, 8192, 8192, ld, b"T", "NT")
    t(8192, 8192, 8192, ld, b"N", "NN")
