"""Which reactors of the bench ensembles are sensitive?  GPU vs CPU oracle, 100 steps, every reactor and zone
checked every 10 steps; writes the reactor indices whose relative error exceeded 1e-7 at any checkpoint
(worst first) to gpurun_out/outliers.json.  oracle/gen_golden.py g10 then runs the Python reference on
exactly those reactors in the build container (run on the GPU box: python tools/outlier_scan.py)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
import wt_oracle as O

THRESH, CAP = 1e-7, 48
out = {"threshold": THRESH, "steps": 100, "every": 10, "ensembles": {}}
for n, N in ((4, 10000), (8, 12500), (20, 10000)):
    steps, every = 100, 10
    cols, bc = wt.make_ensemble(N)
    ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
    s0 = ens.state
    pH, Cl, T, t = s0.pH, s0.chlorine, s0.temperature, s0.time
    worst = np.zeros(N)
    for k in range(steps // every):
        es = ens.step(1.0, n_steps=every)
        pH, Cl, T, t, ost = O.ensemble_step(n, ens.constants, bc, 1.0, every, pH, Cl, T, t, nthreads=32)
        err = np.stack([np.abs(es.pH - pH) / np.abs(pH), np.abs(es.chlorine - Cl) / np.maximum(np.abs(Cl), 1e-300),
                        np.abs(es.temperature - T) / np.abs(T)])
        worst = np.maximum(worst, err.max(axis=(0, 2)))
    idx = np.nonzero(worst > THRESH)[0]
    idx = idx[np.argsort(-worst[idx])][:CAP]
    out["ensembles"][f"n{n}"] = {"reactors_in_ensemble": N, "reactors": [int(i) for i in idx],
                                 "max_rel_err": [float(worst[i]) for i in idx],
                                 "count_over_1e-6": int(np.count_nonzero(worst > 1e-6)),
                                 "count_over_threshold": int(np.count_nonzero(worst > THRESH))}
    print(n, N, out["ensembles"][f"n{n}"]["count_over_threshold"], out["ensembles"][f"n{n}"]["count_over_1e-6"], flush=True)
    ens.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "outliers.json"), "w") as f:
    json.dump(out, f, indent=1)
