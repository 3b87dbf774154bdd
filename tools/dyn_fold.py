"""Fold tools/dyn.sh's PMC passes: counters of the timed launch per wavefront-group and outer step."""
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/dyn/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "step_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    last = {k: v[-1] / 1250.0 / 200.0 for k, v in acc.items()}
    f64 = sum(last.get(k, 0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"))
    print(f"{os.path.basename(d[:-1]):16s} VALU {last.get('SQ_INSTS_VALU', 0):8.1f} (fp64 fma/mul/add {f64:7.1f}) SALU {last.get('SQ_INSTS_SALU', 0):7.1f} "
          f"LDS {last.get('SQ_INSTS_LDS', 0):6.1f} SMEM {last.get('SQ_INSTS_SMEM', 0):6.1f} cycles {4 * last.get('SQ_WAVE_CYCLES', 0):9.0f}")
