#!/bin/bash
# Code-object metadata of the shipped step kernels (register / LDS / spill counts as the loader sees them):
#   bash tools/dump_kernel_meta.sh > profiles/r2/kernel_metadata.txt
set -e
SO=${1:-ics-wt-physicsengine_amd/csrc/libwtphys.so}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $SO
$B/clang-offload-bundler --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co
$B/llvm-readelf --notes $T/dev.co | python3 -c '
import sys, re
txt = sys.stdin.read()
blocks = txt.split("\n  - .agpr_count:")
fields = (".agpr_count", ".vgpr_count", ".sgpr_count", ".sgpr_spill_count", ".vgpr_spill_count", ".group_segment_fixed_size",
          ".private_segment_fixed_size", ".max_flat_workgroup_size", ".wavefront_size")
print("# llvm-readelf --notes of the gfx950 code object embedded in libwtphys.so: the step kernels")
print("# (wt::step_kernel<LV, ROW>: LV = ceil(log2 n) cyclic-reduction levels, ROW = zone count divides 16)")
for b in blocks[1:]:
    b = "    .agpr_count:" + b
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if "step_kernel" not in name: continue
    row = {f: (re.search(re.escape(f) + r":\s+(\S+)", b) or [None, "?"])[1] for f in fields}
    m = re.search(r"step_kernelILi(\d)ELb(\d)", name)
    print(f"wt::step_kernel<{m.group(1)}, {bool(int(m.group(2)))}>  ({name})".replace("True", "true").replace("False", "false"))
    print("    " + "  ".join(f"{f[1:]}={row[f]}" for f in fields))
'
rm -rf $T
