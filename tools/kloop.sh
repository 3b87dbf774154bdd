#!/bin/bash
# static instruction classes of the solver's trip loop (developer metric; compile only, no GPU):  bash tools/kloop.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWT_ONLY_LV3 "$@" -S --cuda-device-only -o tools/scratch/kloop.s ics-wt-physicsengine_amd/csrc/wtphys.hip 2>/dev/null
python3 - <<'EOP'
import re, sys, collections
sys.argv = ["x", "tools/scratch/kloop.s"]
exec(open("tools/asm_loops.py").read().split("print(\"instructions\"")[0])
cands = [(b - a, a, b) for a, b, _ in loops if 3000 < b - a < 9500]
size, a, b = max(cands)
c = collections.Counter(cls(l) for l in ins[a:b + 1])
valu = sum(v for k, v in c.items() if k in ("fp64","v_cmp","accvgpr","v_readlane_b32","v_writelane_b32","cndmask","dpp","v_mov","v_other"))
salu = sum(v for k, v in c.items() if k.startswith("s_"))
print(f"kernel {len(ins)} instrs; trip loop [{a},{b}] {size+1}: VALU {valu} (fp64 {c['fp64']}, non-fp64 {valu - c['fp64']}) SALU {salu} LDS {c['ds']}")
print("   ", dict(c.most_common()))
EOP
