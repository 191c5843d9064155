"""Lab: the reduction stage inside a full eigen_sx solve against the reduction entry point alone, same process, same buffers.
usage: gpu_solve_vs_reduce.py N lda [mf=256]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eigenexa_amd import _lib
n = int(sys.argv[1]); lda = int(sys.argv[2]); mf = int(sys.argv[3]) if len(sys.argv) > 3 else 256
lib = _lib.load(); _lib.check(lib.eigx_init(0), "init")
dev = torch.device("cuda:0"); torch.manual_seed(0)
if os.environ.get("EIGX_SYMGEN"):      # the bench's generator (counter-based R + R^T) instead of torch.rand
    from eigenexa_amd import layout
    R = torch.empty(n, lda, dtype=torch.float64, device=dev)
    R[:, n:] = 0.0
    for c0 in range(0, n, 4096):
        blk = layout.random_symmetric_torch(n, dev, rows=np.arange(n), cols=np.arange(c0, c0 + 4096))
        R[c0:c0 + 4096, :n] = blk.T
        del blk
else:
    R = torch.rand(n, lda, dtype=torch.float64, device=dev)
a = torch.empty_like(R)
z = torch.empty(n, lda, dtype=torch.float64, device=dev)
w = torch.zeros(n, dtype=torch.float64, device=dev)
d = torch.zeros(n, dtype=torch.float64, device=dev); e = torch.zeros(2 * n, dtype=torch.float64, device=dev)
tm = np.zeros(16)
for rep in range(2):
    for mode in (b"A", b"N"):
        a.copy_(R); torch.cuda.synchronize()
        _lib.check(lib.eigx_sx_dev(n, n if mode == b"A" else 0, a.data_ptr(), lda, w.data_ptr(), z.data_ptr(), lda, mf, 128, mode), "sx")
        lib.eigx_get_timers(tm.ctypes.data_as(C.POINTER(C.c_double)))
        print(f"rep {rep} eigen_sx mode {mode.decode()}: total {tm[0]*1e3:.1f} ms, reduction {tm[1]*1e3:.1f}, dc {tm[2]*1e3:.1f}, bt {tm[3]*1e3:.1f}", flush=True)
    a.copy_(R); torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.eigx_band_reduce_dev(n, a.data_ptr(), lda, d.data_ptr(), e.data_ptr(), n, mf, 2), "reduce")
    torch.cuda.synchronize()
    print(f"rep {rep} band_reduce alone: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
