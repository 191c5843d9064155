"""Per-size table of the eigen_h per-column kernels from a rocprofv3 kernel trace (CSV).
usage: herm_trace_table.py <kernel_trace.csv> [N=8192]
The mat-vec of the column with L active rows streams the upper triangle of two planes: 8 L^2 bytes."""
import csv, sys, collections
path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
rows = list(csv.DictReader(open(path)))
hemv = [r for r in rows if "h_hemv_kernel" in r["Kernel_Name"]]
step = [r for r in rows if "h_step_kernel" in r["Kernel_Name"]]
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3   # us
# the launches of one solve come in column order i = n-1 .. 1; use the LAST solve in the trace
hemv = hemv[-(n - 1):]
bk = collections.defaultdict(list)
for idx, r in enumerate(hemv):
    L = n - 1 - idx
    bk[L // 1024].append((L, dur(r)))
print("mat-vec by active size (last solve):")
for b in sorted(bk, reverse=True):
    v = bk[b]; by = sum(8.0 * L * L for L, _ in v); t = sum(d for _, d in v)
    print(f"  L in [{b*1024:5d},{b*1024+1023:5d}]: {len(v):5d} launches  avg {t/len(v):7.2f} us   {by/t*1e-6:5.2f} TB/s")
tot = sum(d for v in bk.values() for _, d in v)
print(f"  total {tot*1e-3:.1f} ms")
st = step[-(n + (n - 1) // 48 + 1):]
ds = [dur(r) for r in st]
print(f"step kernel: {len(ds)} launches, total {sum(ds)*1e-3:.1f} ms, avg {sum(ds)/len(ds):.2f} us, first 1024 avg {sum(ds[:1024])/1024:.2f}, last 1024 avg {sum(ds[-1024:])/1024:.2f}")
