"""MFMA-pipe utilisation of the fp64 GEMM kernels from one rocprofv3 PMC pass (profiles/rNN_gemm_pmc_mfma_busy.csv).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY \
        SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d OUT \
        -- python3 tools/gpu_gemm2_lab.py pmc
    python tools/gemm_pmc.py OUT/.../*_counter_collection.csv profiles/rNN_gemm_pmc_mfma_busy.csv

Derived per launch: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA-busy fraction per SIMD =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) (one v_mfma_f64_16x16x4_f64 keeps a SIMD's matrix pipe busy
for 64 cycles, so 1.0 is the fp64 MFMA peak at the measured clock); TFLOP/s = 512 x SQ_INSTS_VALU_MFMA_MOPS_F64 / duration;
the wait / active fractions are relative to SQ_WAVE_CYCLES."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    d = collections.OrderedDict()
    for r in rows:
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        d[k]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(sys.argv[2], "w") as g:
        g.write("kernel,duration_us,clock_GHz,tflops,mfma_busy_fraction_per_simd,wait_any_frac,wait_inst_frac,"
                "active_inst_frac,mfma_mops_f64\n")
        for (_, name), v in d.items():
            if "gemm" not in name and "Cijk" not in name:
                continue
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            wc = v["SQ_WAVE_CYCLES"]
            short = name.replace("void eigx::(anonymous namespace)::", "").replace(",", ";")[:64]
            g.write(f"{short},{v['dur_us']:.0f},{cyc / v['dur_us'] / 1e3:.2f},"
                    f"{512.0 * v['SQ_INSTS_VALU_MFMA_MOPS_F64'] / v['dur_us'] / 1e6:.1f},"
                    f"{v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):.3f},{v['SQ_WAIT_ANY'] / wc:.2f},"
                    f"{v['SQ_WAIT_INST_ANY'] / wc:.2f},{v['SQ_ACTIVE_INST_ANY'] / wc:.2f},"
                    f"{v['SQ_INSTS_VALU_MFMA_MOPS_F64']:.4g}\n")
    print(open(sys.argv[2]).read())


if __name__ == "__main__":
    main()
