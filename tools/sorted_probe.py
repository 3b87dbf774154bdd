"""What does grouping reactors of similar solver cost into the same wavefront buy? (developer tool; GPU box)
Runs the bench ensemble, measures per-reactor cost, rebuilds the ensemble in cost order from the same state and
times both."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 100
cols, bc = wt.make_ensemble(N)
ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
ens.step(1.0, n_steps=warm, download=False)
cost = np.zeros(N)
for k in range(10):
    ens.step(1.0, n_steps=1, download=False); cost += ens.solver_stats()[:, 0]
st = ens.state
def timed(e, k):
    e.synchronize(); t0 = time.perf_counter(); e.step(1.0, n_steps=k, download=False); e.synchronize(); return time.perf_counter() - t0
t_a = timed(ens, steps)
for label, order in (("same order", np.arange(N)), ("cost order", np.argsort(cost, kind="stable"))):
    c2 = {k: v[order] for k, v in cols.items()}
    e2 = wt.ReactorEnsemble(c2, n_zones=n); e2.set_boundary(np.ascontiguousarray(bc[:, order]))
    e2.set_state(st.pH[order], st.chlorine[order], st.temperature[order], st.time[order])
    t = timed(e2, steps)
    print(f"{label}: {N * n * steps / t:.4g} zone-steps/s ({t / steps * 1e6:.1f} us/step)")
    e2.close()
print(f"original ensemble continuing: {N * n * steps / t_a:.4g}")
