#!/bin/bash
# Static profile of one step kernel of a built library: code-object metadata + instruction-class counts.
#   bash tools/kstat.sh [lib.so] [kernel-name-regex]
set -e
SO=${1:-ics-wt-physicsengine_amd/csrc/libwtphys.so}
PAT=${2:-step_kernelILi3ELb1}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $SO
$B/clang-offload-bundler --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co
$B/llvm-readelf --notes $T/dev.co | python3 -c '
import sys, re
pat = sys.argv[1]
txt = sys.stdin.read()
for b in txt.split("\n  - .agpr_count:")[1:]:
    b = "    .agpr_count:" + b
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if not re.search(pat, name): continue
    f = lambda k: (re.search(re.escape(k) + r":\s+(\S+)", b) or [None, "?"])[1]
    print(name)
    print("  agpr", f(".agpr_count"), "vgpr", f(".vgpr_count"), "sgpr", f(".sgpr_count"), "sgpr_spill", f(".sgpr_spill_count"),
          "vgpr_spill", f(".vgpr_spill_count"), "lds", f(".group_segment_fixed_size"), "scratch", f(".private_segment_fixed_size"))
' "$PAT"
$B/llvm-objdump -d --no-show-raw-insn $T/dev.co | python3 -c '
import sys, re, collections
pat = sys.argv[1]
cur = None; cnt = collections.Counter(); tot = 0
for line in sys.stdin:
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m: cur = m.group(1) if re.search(pat, m.group(1)) and not m.group(1).endswith(".kd") else None; continue
    if not cur: continue
    t = line.split()
    if not t or t[0].endswith(":"): continue
    op = t[0]; tot += 1
    if re.match(r"v_(fma|fmac|mul|add|rcp|div|ldexp|max|min|sqrt|rsq|trunc|floor|frexp|cvt|rndne|fract|cmp|cmpx|cndmask)?.*_f64", op) and "cmp" not in op: cnt["fp64"] += 1
    elif "cmp" in op and op.startswith("v_"): cnt["v_cmp"] += 1
    elif op.startswith("v_accvgpr"): cnt["accvgpr"] += 1
    elif op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): cnt[op] += 1
    elif op.startswith("v_cndmask"): cnt["cndmask"] += 1
    elif "dpp" in op or "dpp" in line: cnt["dpp"] += 1
    elif op.startswith("v_mov"): cnt["v_mov"] += 1
    elif op.startswith("v_"): cnt["v_other"] += 1
    elif op.startswith("s_mov"): cnt["s_mov"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer"): cnt["s_load"] += 1
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"): cnt["s_wait/nop"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): cnt["s_branch"] += 1
    elif op.startswith("s_"): cnt["s_other"] += 1
    elif op.startswith("ds_"): cnt["ds"] += 1
    elif op.startswith("scratch_") or op.startswith("buffer_"): cnt["scratch/buffer"] += 1
    elif op.startswith("global_") or op.startswith("flat_"): cnt["global"] += 1
    else: cnt["other:" + op] += 1
print("  instructions", tot)
for k, v in cnt.most_common(): print(f"    {k:22s} {v}")
' "$PAT"
rm -rf $T
