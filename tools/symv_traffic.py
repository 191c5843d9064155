"""HBM-side traffic of the fused SYMV launches from two rocprofv3 PMC passes (bench.py's roofline.traffic).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex symv_kernel --output-format csv -d OUT_F \
        -- python3 tools/gpu_reduce_time.py 8192 2 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex symv_kernel --output-format csv -d OUT_W \
        -- python3 tools/gpu_reduce_time.py 8192 2 0
    python tools/symv_traffic.py N OUT_F/.../*_counter_collection.csv OUT_W/.../*_counter_collection.csv \
        profiles/rNN_symv_traffic.json profiles/rNN_symv_pmc_nN.csv

Counters are collected in their own passes (no trace domains besides the kernel trace), per launch, in KiB.
gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane
streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as reported.  The k-th SYMV launch of a pentadiagonal
reduction has active size L = N - 2 - 2k and reads 8 L (L + 1) / 2 algorithmic bytes."""
import csv
import json
import sys


def read(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and "symv_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows.sort()
    return rows


def main():
    n = int(sys.argv[1])
    f_rows, w_rows = read(sys.argv[2], "FETCH_SIZE"), read(sys.argv[3], "WRITE_SIZE")
    out_json, out_csv = sys.argv[4], sys.argv[5]
    assert len(f_rows) == len(w_rows) and len(f_rows) > 0, (len(f_rows), len(w_rows))
    nb = 2
    tot_f = tot_w = tot_a = 0.0
    with open(out_csv, "w") as g:
        g.write("step,L,kernel,grid_size,FETCH_SIZE_KB,WRITE_SIZE_KB,algorithmic_bytes\n")
        for k, (fr, wr) in enumerate(zip(f_rows, w_rows)):
            L = n - nb - nb * k
            alg = 8 * L * (L + 1) // 2
            name = fr[1].split("symv_kernel")[1].split(">")[0].replace(" ", "")
            g.write(f"{k},{L},symv_kernel{name}>,{fr[2]},{fr[3]:.6f},{wr[3]:.6f},{alg}\n")
            tot_f += fr[3] * 1024.0
            tot_w += wr[3] * 1024.0
            tot_a += alg
    res = {
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes "
                "(--kernel-include-regex symv_kernel) on tools/gpu_reduce_time.py N 2 0 (one pentadiagonal reduction); "
                "post-processed by tools/symv_traffic.py. gfx950 correction of MI355X_MICROARCH.md (HBM section): "
                "FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes, so it is "
                "doubled; WRITE_SIZE is taken as reported.",
        "n": n,
        "launches": len(f_rows),
        "fetch_bytes_raw": tot_f,
        "fetch_bytes_corrected": 2.0 * tot_f,
        "write_bytes": tot_w,
        "algorithmic_bytes": int(tot_a),
        "traffic_over_algorithmic": (2.0 * tot_f + tot_w) / tot_a,
        "per_launch_csv": out_csv,
    }
    with open(out_json, "w") as g:
        json.dump(res, g, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "note"}))


if __name__ == "__main__":
    main()
