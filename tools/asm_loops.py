"""Static look at a step kernel's assembly (hipcc -S --cuda-device-only): loops (backward branches) and the
instruction classes inside each.  python tools/asm_loops.py file.s [kernel-substring] [min-size] [a:b dump range]"""
import re, sys, collections
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "step_kernelILi3ELb1"; minsize = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = []
labels = {}
for i in range(start, end):
    l = lines[i].split(";")[0].rstrip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = len(ins); continue
    l = l.strip()
    if not l or l.startswith(".") or l.endswith(":"): continue
    ins.append(l)
def cls(l):
    op = l.split()[0]
    if "_f64" in op and "cmp" not in op: return "fp64"
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op in ("v_readlane_b32", "v_writelane_b32"): return op
    if op.startswith("v_cndmask"): return "cndmask"
    if "dpp" in l or "row_" in l or "quad_perm" in l or "wave_sh" in l: return "dpp"
    if op.startswith("v_mov"): return "v_mov"
    if op.startswith("v_"): return "v_other"
    if op.startswith("s_mov"): return "s_mov"
    if op.startswith("s_load"): return "s_load"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "s_wait/nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "s_branch"
    if op.startswith("s_"): return "s_other"
    if op.startswith("ds_"): return "ds"
    if op.startswith("scratch_") or op.startswith("buffer_"): return "scratch"
    if op.startswith("global_") or op.startswith("flat_"): return "global"
    return "other"
loops = []
for i, l in enumerate(ins):
    m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i, m.group(1)))
print("instructions", len(ins))
tot = collections.Counter(cls(l) for l in ins)
print("whole kernel:", dict(tot.most_common()))
for a, b, lab in sorted(loops, key=lambda t: t[0] - t[1]):
    if b - a < minsize: continue
    c = collections.Counter(cls(l) for l in ins[a:b + 1])
    print(f"loop {lab} [{a}, {b}] size {b - a + 1}:", dict(c.most_common()))
if len(sys.argv) > 4:
    a, b = map(int, sys.argv[4].split(":"))
    for i in range(a, b): print(i, ins[i])
