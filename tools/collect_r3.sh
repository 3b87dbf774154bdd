#!/bin/bash
# Round-3 evidence for the dominant kernel, ONE pass over the final build (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command (launch duration agreement with bench.py's HIP events)
#   2. PMC passes, counters in their own runs (--pmc only): HBM traffic for the 500-step AND the driver-style run
#      (the work-item length differs: 32 vs 3 steps), instruction mix, fp64 flop, busy / wait shares; the same for n = 20
#   3. every bench row DESIGN.md quotes
# Output: gpurun_out/r3/ ; tools/summarize_r3.py folds it into profiles/r3/.
set -e
export TMPDIR=/tmp
O=gpurun_out/r3; rm -rf $O; mkdir -p $O
LONG="--no-cpu-baseline --steps 500 --warmup 100"
SHORT="--no-cpu-baseline --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o plain -- python3 bench.py $LONG > $O/bench_under_rocprof.json 2> $O/err.log
echo trace done
pass() {   # pass <dir> <counters> <bench args>
  rocprofv3 --pmc $2 --output-format csv -d $O/$1 -o p -- python3 bench.py $3 > $O/$1.log 2>&1 || { tail -5 $O/$1.log; }
  echo $1 done
}
pass long_fetch "FETCH_SIZE" "$LONG"
pass long_write "WRITE_SIZE" "$LONG"
pass short_fetch "FETCH_SIZE" "$SHORT"
pass short_write "WRITE_SIZE" "$SHORT"
pass long_mix "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM" "$LONG"
pass long_f64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" "$LONG"
pass long_act "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "$LONG"
pass n20_mix "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM" "$LONG --zones 20"
pass n20_f64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" "$LONG --zones 20"
pass n20_fetch "FETCH_SIZE" "$LONG --zones 20"
pass n20_write "WRITE_SIZE" "$LONG --zones 20"
# fold the counter passes into profiles/r3/ first: the bench rows below quote traffic.json / pmc_fp64.json of THIS build
python3 tools/summarize_r3.py --counters-only > $O/summary_counters.log 2>&1 || tail -5 $O/summary_counters.log
# bench rows (no profiler)
python3 bench.py > $O/bench_default.json 2>> $O/err.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2>> $O/err.log
python3 bench.py $LONG --sensors > $O/bench_sensors.json 2>> $O/err.log
python3 bench.py $LONG --plant-io > $O/bench_plantio_scan50.json 2>> $O/err.log
python3 bench.py $LONG --plant-io --chunk 1 > $O/bench_plantio_scan1.json 2>> $O/err.log
python3 bench.py $LONG --reactors 12500 > $O/bench_12500.json 2>> $O/err.log
python3 bench.py $LONG --reactors 12500 --sensors > $O/bench_sensors_12500.json 2>> $O/err.log
python3 bench.py $LONG --zones 4 > $O/bench_n4.json 2>> $O/err.log
python3 bench.py $LONG --zones 16 > $O/bench_n16.json 2>> $O/err.log
python3 bench.py $LONG --zones 20 > $O/bench_n20.json 2>> $O/err.log
python3 bench.py $SHORT --zones 20 > $O/bench_n20_driver_style.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --reactors 100000 --steps 100 --warmup 40 > $O/bench_100k.json 2>> $O/err.log
python3 bench.py $LONG --placement identity > $O/bench_identity_placement.json 2>> $O/err.log
echo done
