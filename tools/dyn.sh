#!/bin/bash
# dynamic instruction counts per wavefront-group-step of scratch builds (one PMC pass each): bash tools/dyn.sh a.so b.so ...
LIST="$*"
/usr/local/graft/bin/gpurun --timeout 1100 -- "export TMPDIR=/tmp; rm -rf gpurun_out/dyn; mkdir -p gpurun_out/dyn; for so in $LIST; do b=\$(basename \$so .so); export WTPHYS_LIB=\$so; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d gpurun_out/dyn/\$b -o p -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 50 > gpurun_out/dyn/\$b.log 2>&1 || tail -3 gpurun_out/dyn/\$b.log; done; python3 tools/dyn_fold.py" 2>&1 | grep -v "^\[gpurun\] s\|merged"
