#!/bin/bash
# registers / spills / scratch of the n = 17..32 kernel (compile only):  bash tools/kmeta5.sh [extra flags]
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWT_ONLY_LV5 "$@" -o tools/scratch/lv5.so ics-wt-physicsengine_amd/csrc/wtphys.hip 2>&1 | grep -E "error" | head -5
bash tools/kstat.sh tools/scratch/lv5.so step_kernelILi5ELb0 | head -3
