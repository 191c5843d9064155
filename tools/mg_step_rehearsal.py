"""Lab: ONE rank of a P-rank grid alone on the GPU (EIGX_LOOPBACK: every peer window is the rank's own memory, it signals
on behalf of every source), so that the complete per-rank kernel sequence of the multi-GPU reduction -- local mat-vec + the
rank's share of the panel dots, kl_kernel (sums to the row owners), ka_kernel over the rank's own rows (x, W to everybody),
the waits, panel gathers, local trailing update -- runs at the TRUE local sizes of that grid on an otherwise idle card and
can be timed.  The numbers computed are meaningless (the other ranks' messages are
copies of this rank's own); xGMI latency and link bandwidth are not part of it.
usage: mg_step_rehearsal.py P rank N [band=2] [mf=256] [PxxPy]        (run under rocprofv3 --kernel-trace for durations)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EIGX_LOOPBACK"] = "1"
import numpy as np
import torch
from eigenexa_amd import _lib, layout

P, rank, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
band = int(sys.argv[4]) if len(sys.argv) > 4 else 2
mf = int(sys.argv[5]) if len(sys.argv) > 5 else 256
lib = _lib.load()
if len(sys.argv) > 6:
    px_, py_ = (int(v) for v in sys.argv[6].split("x"))
    lib.eigx_set_grid_dims(px_, py_)
uid = C.create_string_buffer(bytes(range(128)), 128)
_lib.check(lib.eigx_init_multi(0, rank, P, uid, b"C"), "eigx_init_multi (loopback)")
for kv in filter(None, os.environ.get("EIGX_TUNE", "").split(",")):
    lib.eigx_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
p_ = C.c_int(); xp = C.c_int(); yp = C.c_int(); i_ = C.c_int(); xi = C.c_int(); yi = C.c_int()
lib.eigx_get_procs(C.byref(p_), C.byref(xp), C.byref(yp)); lib.eigx_get_id(C.byref(i_), C.byref(xi), C.byref(yi))
Px, Py, px, py = xp.value, yp.value, xi.value - 1, yi.value - 1
dev = torch.device("cuda:0")
rows = np.arange(px, n, Px); cols = np.arange(py, n, Py)
# leading dimension of the local block: what eigen_get_matdims recommends for this grid (EIGX_NX=old: the pre-round-3 choice)
nx_c, ny_c = C.c_int(), C.c_int()
lib.eigx_matdims_for_grid(n, Px, Py, mf, 128, b"O", C.byref(nx_c), C.byref(ny_c))
nx = (len(rows) + 63) // 64 * 64 + 34 if os.environ.get("EIGX_NX") == "old" else max(nx_c.value, len(rows))
print(f"local block {len(rows)} x {len(cols)}, leading dimension {nx}", flush=True)
a = torch.zeros(len(cols) + 8, nx, dtype=torch.float64, device=dev)
for c0 in range(0, len(cols), 1024):          # chunked: the generator makes (rows x chunk) temporaries
    blk = layout.random_symmetric_torch(n, dev, rows=rows, cols=cols[c0:c0 + 1024])
    a[c0:c0 + blk.shape[1], :len(rows)] = blk.T
    del blk
d = torch.zeros(n, dtype=torch.float64, device=dev)
e = torch.zeros(2 * n, dtype=torch.float64, device=dev)
a0 = a.clone()
for rep in range(2):                           # first pass allocates the workspace
    a.copy_(a0)
    lib.eigx_profile(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.eigx_band_reduce_dev(n, a.data_ptr(), nx, d.data_ptr(), e.data_ptr(), n, mf, band)
    dt = time.perf_counter() - t0
    kinds = np.zeros(18)
    lib.eigx_profile_read_kinds(kinds.ctypes.data_as(C.POINTER(C.c_double)), 6)
    lib.eigx_profile(0)
    steps = n // band
    us = lambda k: kinds[3 * k + 2] / max(kinds[3 * k], 1) * 1e6
    print(f"rank {rank} ({px},{py}) of {Px}x{Py}, N={n} band={band} mf={mf} rep {rep}: rc {rc}, reduction {dt*1e3:.1f} ms = {dt/steps*1e6:.2f} us per step "
          f"over {steps} steps; sampled averages over the whole reduction (HIP events, us): mat-vec + K_P {us(0):.2f}, "
          f"kl (sums to owners) {us(2):.2f}, wait Y {us(3):.2f}, ka (own rows) {us(4):.2f}, wait X {us(5):.2f}, "
          f"trailing update {us(1):.1f} x {int(kinds[3])}; d, e finite: {bool(torch.isfinite(d).all() and torch.isfinite(e).all())}", flush=True)
