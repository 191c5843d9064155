"""Timeline of the D&C stage of the LAST solve in a rocprofv3 kernel trace (csv): start (us after the leaf kernel), duration,
queue, short kernel name; gaps > 20 us on the union of all queues are marked.  Usage: dc_timeline.py <kernel_trace.csv>"""
import csv, sys, re

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
leafs = [i for i, r in enumerate(rows) if "jacobi_leaf" in r[2]]
perms = [i for i, r in enumerate(rows) if "final_permute" in r[2]]
i0, i1 = leafs[-1], [p for p in perms if p > leafs[-1]][0]
t0 = rows[i0][0]
busy_end = t0
print(f"D&C window of the last solve: {(rows[i1][1]-t0)/1e3:.1f} us, {i1-i0+1} kernels")
for s, e, name, q in rows[i0:i1 + 1]:
    m = re.search(r"(\w+)(<[^>]*>)?\(", name)
    short = (m.group(1) + (m.group(2) or "")) if m else name[:40]
    gap = (s - busy_end) / 1e3
    mark = f"   <-- idle {gap:.0f} us" if gap > 20 else ""
    print(f"{(s-t0)/1e3:10.1f} {(e-s)/1e3:9.1f} q{q:>3} {short}{mark}")
    busy_end = max(busy_end, e)
if len(sys.argv) > 2 and sys.argv[2] == "bt":   # what follows the D&C in the same solve (the back-transformation)
    print("---- after the D&C ----")
    t1 = rows[i1][1]
    for s, e, name, q in rows[i1 + 1:]:
        if "at::native" in name or "Cijk" in name:
            break
        m = re.search(r"(\w+)(<[^>]*>)?\(", name)
        short = (m.group(1) + (m.group(2) or "")) if m else name[:40]
        print(f"{(s-t1)/1e3:10.1f} {(e-s)/1e3:9.1f} q{q:>3} {short}")
