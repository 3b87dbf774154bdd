"""Why are short runs slow?  Per-call wall time of consecutive 5-step calls on one ensemble (transient + clock ramp),
then the same on a fresh ensemble while the GPU is already warm (transient only)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
N, n = 10000, 8
cols, bc = wt.make_ensemble(N)
def series(label, calls=40, k=5):
    ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
    out = []
    for c in range(calls):
        ens.synchronize(); t0 = time.perf_counter()
        ens.step(1.0, n_steps=k, download=False); ens.synchronize()
        out.append((time.perf_counter() - t0) * 1e3 / k)
    ens.close()
    print(label, "ms/step per 5-step call:", " ".join(f"{x:.3f}" for x in out), flush=True)
series("cold GPU, fresh ensemble ")
series("warm GPU, fresh ensemble ")
time.sleep(2.0)
series("after 2 s idle, fresh    ")
