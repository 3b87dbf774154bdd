#!/bin/bash
# A/B timing of scratch builds on one box, interleaved: bash tools/ab.sh a.so b.so [repeats]
A=$1; B=$2; R=${3:-3}
/usr/local/graft/bin/gpurun --timeout 900 -- "for i in \$(seq $R); do for so in $A $B; do WTPHYS_LIB=\$so python bench.py --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(\"'\$so'\", \"500:\", \"%.4g\" % d[\"value\"])'; WTPHYS_LIB=\$so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(\"'\$so'\", \"20:\", \"%.4g\" % d[\"value\"])'; done; done" 2>&1 | grep -E "\.so" | sort | awk '{k=$1" "$2; s[k]=s[k]" "$3} END {for (k in s) print k, s[k]}' | sort
