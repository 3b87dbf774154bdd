"""How much wavefront time would binning reactors by solver cost save? (developer tool)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
R = 64 // n
cols, bc = wt.make_ensemble(N)
ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
ens.step(1.0, n_steps=100, download=False)
hist = []
for k in range(60):
    ens.step(1.0, n_steps=1, fused=False, download=False)
    hist.append(ens.solver_stats()[:, 0].copy())      # nfev of that outer step
hist = np.array(hist, dtype=np.float64)               # (60, N)
a, b = hist[:30].mean(0), hist[30:].mean(0)
print("nfev per step: mean %.2f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (hist.mean(), np.median(hist), np.percentile(hist, 90), np.percentile(hist, 99), hist.max()))
print("temporal correlation of per-reactor mean cost (first 30 vs next 30 steps): %.3f" % np.corrcoef(a, b)[0, 1])
def wave_cost(order, h):
    m = (N // R) * R
    return h[:, order[:m]].reshape(h.shape[0], -1, R).max(2).sum(1).mean()
ident = np.arange(N)
srt = np.argsort(a, kind="stable")
print("sum over waves of max nfev (proxy for wave time), per step, evaluated on the LAST 30 steps:")
print("  original order  %.0f" % wave_cost(ident, hist[30:]))
print("  sorted by first-30-step cost %.0f" % wave_cost(srt, hist[30:]))
print("  ideal (mean cost, no divergence) %.0f" % (hist[30:].mean() * (N // R)))
