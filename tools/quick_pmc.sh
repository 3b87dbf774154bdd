#!/bin/bash
# quick instruction-mix counters of the step kernel (developer tool): bash tools/quick_pmc.sh [bench args]
export TMPDIR=/tmp
O=gpurun_out/qpmc; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 50 "$@" > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python3 - ${WT_GROUPS:-1250} <<'EOP'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/qpmc/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "step_kernel" in row["Kernel_Name"] or "triad_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
last = {k: v[-1] for k, v in acc.items()}   # the timed launch
import sys
waves = float(sys.argv[1]) if len(sys.argv) > 1 else 1250.0; steps = 200.0
for k in sorted(last):
    print(f"{k:28s} {last[k]:.4e}  per group-step {last[k]/waves/steps:10.1f}")
EOP
