#!/bin/bash
# usage: tools/bench_grid.sh "<bench args>" ... ; prints one summary line per configuration
for args in "$@"; do
  python bench.py --no-cpu-baseline $args > /tmp/_b.json 2>/dev/null || { echo "FAILED: $args"; continue; }
  python - "$args" <<'PY'
import json, sys
d = json.load(open("/tmp/_b.json")); r = d["roofline"]
print(f"{sys.argv[1]:40s} {d['value']:.3e}  launch avg {r['avg_launch_us']:.0f} max {r['max_launch_us']:.0f} us  in flight {r['launches_in_flight']:.2f}")
PY
done
