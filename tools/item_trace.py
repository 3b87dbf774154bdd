"""Timeline of the work items of one queue-schedule launch: how busy the workers are, when they start / retire,
what the slowest group's chain costs (developer tool; GPU box)."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
nat = importlib.import_module("ics-wt-physicsengine_amd.core._native")
n, N = 8, int(sys.argv[2]) if len(sys.argv) > 2 else 10000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols, bc = wt.make_ensemble(N)
ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
ens.step(1.0, n_steps=5, download=False); ens.synchronize()
cap = 200000; cnt = C.c_int(0)
nat.check(nat.lib().wt_ensemble_item_trace(ens._h, None, cap, C.byref(cnt)))
ens.timer_start(); ens.step(1.0, n_steps=steps, download=False); ms = ens.timer_stop()
buf = np.zeros((cap, 5), dtype=np.int64)
nat.check(nat.lib().wt_ensemble_item_trace(ens._h, buf.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(cnt)))
tr = buf[:cnt.value]
t0 = tr[:, 3].min(); st = (tr[:, 3] - t0) / 100.0; en = (tr[:, 4] - t0) / 100.0   # us
span = en.max()
print(f"launch {ms*1e3:.0f} us (HIP events), items {len(tr)}, span of items {span:.0f} us, workers seen {len(np.unique(tr[:,0]))}")
busy = np.zeros(tr[:, 0].max() + 1); first = np.full_like(busy, 1e18); last = np.zeros_like(busy)
np.add.at(busy, tr[:, 0], en - st); np.minimum.at(first, tr[:, 0], st); np.maximum.at(last, tr[:, 0], en)
print(f"worker busy us: mean {busy.mean():.0f} min {busy.min():.0f} max {busy.max():.0f}; first start us: p50 {np.median(first):.1f} max {first.max():.1f}; "
      f"retire us: p10 {np.percentile(last,10):.0f} p50 {np.median(last):.0f} p90 {np.percentile(last,90):.0f} max {last.max():.0f}")
print(f"utilisation over the span: {busy.sum() / (span * len(busy)) * 100:.1f} %")
g = tr[:, 1]; gb = np.zeros(g.max() + 1); np.add.at(gb, g, en - st); gend = np.zeros_like(gb); np.maximum.at(gend, g, en)
print(f"group compute us: mean {gb.mean():.0f} p90 {np.percentile(gb,90):.0f} p99 {np.percentile(gb,99):.0f} max {gb.max():.0f}; group finish us: p50 {np.median(gend):.0f} p99 {np.percentile(gend,99):.0f} max {gend.max():.0f}")
worst = np.argsort(-gend)[:5]
for w in worst:
    m = tr[:, 1] == w; o = np.argsort(st[m])
    print(f" group {w}: compute {gb[w]:.0f} finish {gend[w]:.0f}; items (start,dur,worker):", [(round(float(a)), round(float(b - a)), int(c)) for a, b, c in zip(st[m][o], en[m][o], tr[m][o, 0])][:12])
ens.close()
