"""Static estimate of exposed LDS / scalar-load waits in the trip loop (developer tool).
   python3 tools/wait_model.py tools/scratch/kloop.s [LBB prefix]   -- per basic block: waits and the stall each one is
   expected to cost with a lone wavefront (issue 5.3 cycles per instruction, LDS / SMEM round trip 64 cycles: tools/abn.sh
   on the WT_PAD=7..9 builds)."""
import re, sys, collections
path = sys.argv[1]; LAT = 64.0; ISSUE = 5.3
lines = open(path).read().split('\n')
idx = [i for i, l in enumerate(lines) if 'Header=BB11_79' in l or l.startswith('.LBB11_79:')]
lo, hi = min(idx), max(idx) + 400
blocks = []; cur = None
for i in range(lo, hi):
    l = lines[i]
    if l.startswith('.LBB') or l.startswith('; %bb.'):
        if i > max(idx) and l.startswith('.LBB') and 'Header=BB11_79' not in l: break
        cur = {'name': l.split()[0] if l.startswith('.LBB') else l.split()[1], 'line': i + 1, 'n': 0, 'out': [], 'stall': 0.0, 'waits': []}
        blocks.append(cur); continue
    t = l.strip()
    if not t or t[0] in ';.' or cur is None: continue
    op = t.split()[0]
    cur['n'] += 1
    if op.startswith('s_load') or (op.startswith('ds_') and not op.startswith('ds_nop')):
        cur['out'].append((cur['n'], op))
    m = re.match(r's_waitcnt.*lgkmcnt\((\d+)\)', t)
    if m:
        keep = int(m.group(1))
        must = cur['out'][:len(cur['out']) - keep] if keep else cur['out'][:]
        if must:
            n_issue, o = must[-1]
            st = max(0.0, LAT - (cur['n'] - n_issue) * ISSUE)
            cur['stall'] += st; cur['waits'].append((cur['n'], o, round(st)))
        else:
            cur['waits'].append((cur['n'], 'cross-block', -1))
        cur['out'] = cur['out'][len(must):]
tot = 0
for b in blocks:
    if b['waits']:
        print(f"{b['name']:12s} L{b['line']:6d} n={b['n']:4d} stall~{b['stall']:5.0f}  " + ' '.join(f"@{w[0]}:{w[1].replace('s_load_dword','sl').replace('ds_read','dr').replace('ds_write','dw')}:{w[2]}" for w in b['waits']))
        tot += b['stall']
print('static total', tot)
