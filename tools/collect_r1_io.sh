#!/bin/bash
# Round-1 measurements of the widened rows (sensor suite / plant I/O): run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
O=gpurun_out/r1io; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/err.log
python bench.py --sensors --no-cpu-baseline > $O/bench_sensors.json 2>> $O/err.log
python bench.py --plant-io --no-cpu-baseline > $O/bench_plantio_chunk25.json 2>> $O/err.log
python bench.py --plant-io --chunk 1 --no-cpu-baseline --steps 300 --warmup 50 > $O/bench_plantio_chunk1.json 2>> $O/err.log
python bench.py --chunk 1 --no-cpu-baseline --steps 300 --warmup 50 > $O/bench_plain_chunk1.json 2>> $O/err.log
python bench.py --sensors --reactors 12500 --no-cpu-baseline > $O/bench_sensors_12500.json 2>> $O/err.log
echo benches done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o cfg5 -- python3 bench.py --plant-io --no-cpu-baseline --steps 200 --warmup 50 > $O/trace.log 2>&1
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --plant-io --no-cpu-baseline --steps 100 --warmup 25 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --plant-io --no-cpu-baseline --steps 100 --warmup 25 > $O/pmc_write.log 2>&1
echo pmc done
find $O -name "*.csv" | head -20
