"""In-kernel bandwidth of the fused mat-vec over the LAST solve of a rocprofv3 kernel trace (CSV) of bench.py --size N.
usage: symv_last_solve.py <kernel_trace.csv> N [solves_in_trace]
The mat-vec of a step with L active rows streams the upper triangle once: 8 * L^2 / 2 bytes per launch (2 columns per step)."""
import csv, sys, collections
path = sys.argv[1]; n = int(sys.argv[2]); nsolve = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = [r for r in csv.DictReader(open(path)) if "symv_kernel" in r["Kernel_Name"]]
per = len(rows) // nsolve
last = rows[-per:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
tot = sum(dur(r) for r in last)
alg = 2.0 / 3.0 * float(n) ** 3          # sum over steps of 4 L^2 bytes, L = n, n-2, ...
print(f"{len(rows)} mat-vec launches in the trace, {per} per solve; last solve: {tot*1e3:.1f} ms in the kernel, "
      f"{alg/tot*1e-12:.2f} TB/s = {alg/tot/8e12:.3f} of 8 TB/s")
for s in range(nsolve):
    seg = rows[s * per:(s + 1) * per]
    t = sum(dur(r) for r in seg)
    print(f"  solve {s}: {t*1e3:.1f} ms in the kernel, {alg/t*1e-12:.2f} TB/s")
bk = collections.defaultdict(lambda: [0.0, 0.0])
for idx, r in enumerate(last):
    L = n - 2 * idx
    b = L // 4096
    bk[b][0] += 4.0 * L * L; bk[b][1] += dur(r)
for b in sorted(bk, reverse=True):
    print(f"  L in [{b*4096:6d},{b*4096+4095:6d}]: {bk[b][0]/bk[b][1]*1e-12:.2f} TB/s")
