import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
import wt_oracle as O
import test_gpu_parity as T
for n, dt, steps in [(4, 1.0, 6), (8, 0.1, 6), (5, 10.0, 4), (8, 30.0, 3), (20, 100.0, 2)]:
    N = 1500
    cols, bc = T._random_edge_ensemble(wt, N, seed=1000 * n + int(dt * 10))
    ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc); ens.set_step_limit(300); O.set_step_limit(300)
    s0 = ens.state
    es = ens.step(dt, n_steps=steps)
    pH, Cl, Tt, t, ost = O.ensemble_step(n, ens.constants, bc, dt, steps, s0.pH, s0.chlorine, s0.temperature, s0.time, nthreads=16)
    O.set_step_limit(0)
    lim = ((es.status | ost.astype(np.uint32)) & 128) != 0
    ok = ((ost & (64 | 2)) == 0) & ~lim
    got = np.stack([es.pH, es.chlorine, es.temperature])[:, ok]; ref = np.stack([pH, Cl, Tt])[:, ok]
    ad = np.abs(got - ref); rel = ad / np.maximum(np.abs(ref), 1e-30)
    mix = ad / (1e-6 * np.abs(ref) + 1e-8)
    print(n, dt, "lim", lim.sum(), "status mismatch", int((es.status[~lim] != ost[~lim]).sum()), "max mixed-tol ratio %.3g" % mix.max(),
          "p99.9 %.3g" % np.percentile(mix, 99.9), "frac rel<1e-6 %.5f" % np.mean(rel[ref != 0] < 1e-6), "worst sample", np.unravel_index(mix.argmax(), mix.shape))
    ens.close()
