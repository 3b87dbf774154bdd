#!/bin/bash
# static instruction classes per stamped section of the solver loop (developer metric; compile only):  bash tools/ksect.sh
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWT_ONLY_LV3 -DWT_STAMPS "$@" -S --cuda-device-only -o tools/scratch/lv3_stamps.s ics-wt-physicsengine_amd/csrc/wtphys.hip 2>/dev/null
python3 - <<'EOP'
import re, collections, sys
sys.argv = ["x", "tools/scratch/lv3_stamps.s"]
exec(open("tools/asm_loops.py").read().split("loops = []")[0])
pos = [i for i, l in enumerate(ins) if l.startswith("s_memtime")]
for k, (a, b) in enumerate(zip(pos[:-1], pos[1:])):
    c = collections.Counter(cls(l) for l in ins[a:b])
    valu = sum(v for kk, v in c.items() if kk in ("fp64","v_cmp","accvgpr","v_readlane_b32","v_writelane_b32","cndmask","dpp","v_mov","v_other"))
    print(f"[{a},{b}) size {b-a} VALU {valu} fp64 {c['fp64']}:", {kk: v for kk, v in c.most_common() if kk != 'fp64'})
EOP
