#!/bin/bash
# interleaved timing of any number of scratch builds on one box: bash tools/abn.sh <repeats> a.so b.so ...   (500-step run only)
R=$1; shift
/usr/local/graft/bin/gpurun --timeout 1100 -- "for i in \$(seq $R); do for so in $*; do WTPHYS_LIB=\$so python bench.py --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(\"'\$so'\", \"%.4g\" % d[\"value\"], repr(d[\"state_checksum\"]))'; done; done" 2>&1 | grep -E "\.so" | sort | awk '{k=$1; s[k]=s[k]" "$2; c[k]=$3} END {for (k in s) print k, s[k], c[k]}' | sort
