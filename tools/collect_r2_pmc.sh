#!/bin/bash
# Round-2 evidence for the dominant kernel (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command (launch duration agreement with bench.py's HIP events)
#   2. PMC passes, counters in their own runs (--pmc only): HBM traffic, instruction mix, fp64 flop, busy / wait shares
# Output: gpurun_out/r2pmc/ ; tools/summarize_r2_pmc.py folds it into profiles/r2/.
set -e
export TMPDIR=/tmp
O=gpurun_out/r2pmc; rm -rf $O; mkdir -p $O
ARGS="--no-cpu-baseline --steps 500 --warmup 100"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o plain -- python3 bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/err.log
python3 bench.py > $O/bench_default.json 2>> $O/err.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2>> $O/err.log
echo trace done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -o p -- python3 bench.py $ARGS > $O/pmc$i.log 2>&1 || { tail -5 $O/pmc$i.log; }
  echo pass $i done
done
# sensors + plant I/O row (config 5 per-GPU shape): trace only
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_io -o plain -- python3 bench.py $ARGS --plant-io --chunk 1 > $O/bench_plantio_scan1.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --sensors > $O/bench_sensors.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --plant-io > $O/bench_plantio_scan50.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --plant-io --chunk 1 > $O/bench_plantio_scan1_plain.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --reactors 12500 --sensors > $O/bench_sensors_12500.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --reactors 12500 > $O/bench_12500.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --zones 4 > $O/bench_n4.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --zones 20 > $O/bench_n20.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --reactors 100000 --steps 100 --warmup 40 > $O/bench_100k.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --streams 4 > $O/bench_streams4.json 2>> $O/err.log
echo done
