#!/bin/bash
# Round-1 evidence for the dominant kernel (run on the GPU box from the repo root):
# kernel trace + launch agreement, then PMC passes (counters in their own runs, --pmc only).
set -e
export TMPDIR=/tmp
O=gpurun_out/r1pmc; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o plain -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/err.log
python bench.py > $O/bench_default.json 2>> $O/err.log
echo trace done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -o p -- python3 bench.py --no-cpu-baseline --steps 500 --warmup 100 > $O/pmc$i.log 2>&1
  echo pass $i done
done
