"""Timing only: band reduction (eigx_band_reduce_dev) of a random symmetric matrix generated on the GPU.
usage: gpu_reduce_time.py N [band=2] [reps=1]   (library selectable with EIGX_LIB for A/B runs; EIGX_MF = panel width)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigenexa_amd import _lib

n = int(sys.argv[1]); band = int(sys.argv[2]) if len(sys.argv) > 2 else 2; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = _lib.load()
_lib.check(lib.eigx_init(0), "eigx_init")
if os.environ.get("EIGX_PROFILE_EVERY"): lib.eigx_profile(int(os.environ["EIGX_PROFILE_EVERY"]))   # the bench's per-launch events
if os.environ.get("EIGX_T128"): lib.eigx_tune(3, int(os.environ["EIGX_T128"]))
if os.environ.get("EIGX_T256"): lib.eigx_tune(4, int(os.environ["EIGX_T256"]))
if os.environ.get("EIGX_NT"): lib.eigx_tune(5, int(os.environ["EIGX_NT"]))
dev = torch.device("cuda:0")
torch.manual_seed(0)
lda = int(os.environ.get("EIGX_LDA", n + (n & 1) + 2 + 30))   # even, not a multiple of a large power of two
R = torch.rand(n, lda, dtype=torch.float64, device=dev)
d = torch.zeros(n, dtype=torch.float64, device=dev)
e = torch.zeros(band * n, dtype=torch.float64, device=dev)
# EIGX_VARIANTS="3=9000,4=40000;3=40000;8=1;..." : one eigx_tune setting list per rep, cycled, all on the SAME buffer
# (the time of a reduction depends on the buffer it runs on -- see DESIGN.md -- so A/B runs must not change buffers)
variants = [v for v in os.environ.get("EIGX_VARIANTS", "").split(";") if v]
defaults = {}
a_fixed = torch.empty_like(R) if variants else None
for rep in range(reps + 1):
    if variants:
        a = a_fixed
        a.copy_(R)
        for k_, v_ in defaults.items():
            lib.eigx_tune(k_, v_)
        cur = variants[rep % len(variants)]
        for kv in cur.split(","):
            if "=" in kv:
                k_, v_ = (int(t) for t in kv.split("="))
                old = lib.eigx_tune(k_, v_)
                defaults.setdefault(k_, old)
        os.environ["EIGX_LIB"] = "variant[" + cur + "]"
    else:
        a = R.clone()
    a[:, :n] = a[:, :n] + a[:, :n].T
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.eigx_band_reduce_dev(n, a.data_ptr(), lda, d.data_ptr(), e.data_ptr(), n, int(os.environ.get('EIGX_MF', '128')), band), "reduce")
    dt = time.perf_counter() - t0
    print(f"{os.environ.get('EIGX_LIB', 'default')[-24:]} t128={os.environ.get('EIGX_T128','-')} t256={os.environ.get('EIGX_T256','-')} nt={os.environ.get('EIGX_NT','-')} mf={os.environ.get('EIGX_MF','128')} n={n} band={band} rep {rep}: {dt*1e3:.1f} ms  "
          f"({4.0/3.0*n**3/dt/1e12:.2f} TFLOP/s of 4/3 n^3; d[0]={d[0].item():.6f})", flush=True)
