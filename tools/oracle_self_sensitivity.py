"""How far do two rounding-level-different CPU executions of the SAME algorithm drift apart?
Oracle with scipy's dense LU vs oracle with the tridiagonal solve, bench ensemble, 100 steps.
(CPU only; quantifies the intrinsic sensitivity of the reference's adaptive solve.)"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
import wt_oracle as O
res = {}
for n in (4, 8, 20):
    N, steps, every = 10000, 100, 10
    cols, bc = wt.make_ensemble(N)
    d = wt.ReactorConfiguration()
    full = {k: np.broadcast_to(np.asarray(cols.get(k, getattr(d, k))), (N,)).copy()
            for k in ("volume", "height", "diameter", "flow_rate", "impeller_speed", "impeller_diameter",
                      "total_carbonate", "temperature", "enable_thermal_stratification")}
    par = wt.params.derive_constants(full, n)
    shape = (N, n)
    A = [np.broadcast_to(cols[k][:, None], shape).copy() for k in ("initial_pH", "initial_chlorine", "temperature")] + [np.zeros(N)]
    B = [x.copy() for x in A]
    allerr = []; worst = []
    for k in range(steps // every):
        O.set_linsolve(0); A = list(O.ensemble_step(n, par, bc, 1.0, every, *A, nthreads=8))[:4]
        O.set_linsolve(1); B = list(O.ensemble_step(n, par, bc, 1.0, every, *B, nthreads=8))[:4]
        err = np.stack([np.abs(A[i] - B[i]) / np.maximum(np.abs(A[i]), 1e-300) for i in range(3)])
        allerr.append(err.reshape(-1)); worst.append(float(err.max()))
    O.set_linsolve(0)
    allerr = np.concatenate(allerr)
    res[f"10000x{n}"] = {"max_rel_err_per_checkpoint": worst, "fraction_within_1e-6": float(np.mean(allerr <= 1e-6)),
                         "fraction_within_1e-9": float(np.mean(allerr <= 1e-9)), "p99.99": float(np.percentile(allerr, 99.99)),
                         "p100": float(allerr.max())}
    print(n, res[f"10000x{n}"], flush=True)
json.dump(res, open(os.path.join(ROOT, "profiles", "r1", "oracle_self_sensitivity.json"), "w"), indent=1)
