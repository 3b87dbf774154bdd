#!/bin/bash
# interleaved timing of scratch builds on the driver's short run: bash tools/ab20.sh <repeats> a.so b.so ...
R=$1; shift
/usr/local/graft/bin/gpurun --timeout 1100 -- "for i in \$(seq $R); do for so in $*; do WTPHYS_LIB=\$so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(\"'\$so'\", \"%.4g\" % d[\"value\"])'; done; done" 2>&1 | grep -E "\.so" | sort | awk '{k=$1; s[k]=s[k]" "$2} END {for (k in s) print k, s[k]}' | sort
