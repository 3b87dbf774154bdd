#!/bin/bash
# developer loop: bits + speed of a scratch build (n = 8 only).  bash tools/try.sh tools/scratch/lv3.so [label]
SO=${1:-tools/scratch/lv3.so}; L=${2:-try}
/usr/local/graft/bin/gpurun --timeout 600 -- "export WTPHYS_LIB=$SO; python tools/bits_check.py quick > gpurun_out/$L.txt 2>&1; python bench.py --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(\"500-step\", \"%.4g\" % d[\"value\"], d[\"state_checksum\"])' >> gpurun_out/$L.txt; python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(\"20-step\", \"%.4g\" % d[\"value\"], d[\"state_checksum\"])' >> gpurun_out/$L.txt; cat gpurun_out/$L.txt" 2>&1 | grep -v "^\[gpurun\] s\|merged"
