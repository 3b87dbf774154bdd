"""Parity report: HIP path vs the reference's golden trajectories and vs the CPU oracle on
the full-size bench ensembles, and vs the reference itself on the reactors where those two differ most (g10).
Writes gpurun_out/parity_report.json (run on the GPU box; copied to profiles/r3/)."""
import glob, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
import wt_oracle as O
from conftest import cfg_columns

out = {"tolerance_north_star": 1e-6, "golden_trajectories": {}, "bench_ensemble_vs_oracle": {}}
PCT = (50, 90, 99, 99.9, 99.99, 100)

def pct(err):
    return {f"p{p}": float(np.percentile(err, p)) for p in PCT}

# 1. golden trajectories (reference Python, every step)
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "g3_traj_*.npz"))):
    g = np.load(path)
    name = os.path.basename(path)[8:-4]
    n = g["traj"].shape[2]
    cols = cfg_columns(g["cfg"], g["cfg_fields"])
    ens = wt.ReactorEnsemble(cols, n_zones=n)
    ens.set_boundary(np.ascontiguousarray(g["bc"][:, None]))
    ens.set_schedule(1, 1)
    traj, stats, dt = g["traj"], g["stats"], float(g["dt"])
    nst = traj.shape[0] - 1
    errs = np.empty((nst, 3, n)); mism = []
    for k in range(nst):
        es = ens.step(dt, n_steps=1)
        got = np.stack([es.pH[0], es.chlorine[0], es.temperature[0]])
        errs[k] = np.abs(got - traj[k + 1]) / np.abs(traj[k + 1])
        if tuple(ens.solver_stats()[0][:4]) != tuple(stats[k][:4]):
            mism.append(k)
    out["golden_trajectories"][name] = {
        "steps": nst, "max_rel_pH": float(errs[:, 0].max()), "max_rel_Cl": float(errs[:, 1].max()),
        "max_rel_T": float(errs[:, 2].max()), "steps_with_different_solver_counters": mism,
        "internal_steps_per_outer_step_max": int(stats[:, 3].max())}
    ens.close()

# 2. bench ensembles at full size vs the oracle, 100 steps, every reactor and zone
for n in (4, 8, 20):
    N, steps, every = 10000, 100, 10
    cols, bc = wt.make_ensemble(N)
    ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
    s0 = ens.state
    pH, Cl, T, t = s0.pH, s0.chlorine, s0.temperature, s0.time
    worst = []; allerr = []
    counters_equal = 0; counters_total = 0
    for k in range(steps // every):
        es = ens.step(1.0, n_steps=every)
        pH, Cl, T, t, ost = O.ensemble_step(n, ens.constants, bc, 1.0, every, pH, Cl, T, t, nthreads=32)
        err = np.stack([np.abs(es.pH - pH) / np.abs(pH), np.abs(es.chlorine - Cl) / np.maximum(np.abs(Cl), 1e-300),
                        np.abs(es.temperature - T) / np.abs(T)])
        allerr.append(err.reshape(-1)); worst.append(float(err.max()))
        assert np.array_equal(es.status != 0, ost != 0)
    allerr = np.concatenate(allerr)
    out["bench_ensemble_vs_oracle"][f"10000x{n}"] = {
        "steps": steps, "compared_every": every, "samples": int(allerr.size), "max_rel_err_per_checkpoint": worst,
        "fraction_within_1e-6": float(np.mean(allerr <= 1e-6)), "fraction_within_1e-9": float(np.mean(allerr <= 1e-9)),
        **pct(allerr)}
    ens.close()

# 3. the outlier reactors of the bench ensembles (GPU vs oracle > 1e-7) against 100 steps of the Python reference
#    itself (tests/golden/g10_outliers_n*.npz), with the oracle's own distance to the reference beside it
from test_oracle_golden import outlier_errors
out["outlier_reactors_vs_reference"] = {}
for n in (4, 8, 20):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"g10_outliers_n{n}.npz"))
    R, every, steps = g["reactors"], int(g["every"]), int(g["steps"])
    cols, bc = wt.make_ensemble(int(R.max()) + 1)
    ens = wt.ReactorEnsemble({k: v[R] for k, v in cols.items()}, n_zones=n)
    ens.set_boundary(np.ascontiguousarray(bc[:, R]))
    S = len(R); e_gpu = np.zeros(S); same = 0
    for k in range(steps // every):
        es = ens.step(1.0, n_steps=every)
        snap = g["snaps"][:, k]
        got = np.stack([es.pH, es.chlorine, es.temperature], axis=1)
        e_gpu = np.maximum(e_gpu, np.max(np.abs(got - snap) / np.abs(snap), axis=(1, 2)))
        same += int(np.count_nonzero(np.all(ens.solver_stats()[:, :4] == g["stats"][:, (k + 1) * every - 1, :4], axis=1)))
    ens.close()
    e_lu, e_tri = outlier_errors(wt, O, n, 0), outlier_errors(wt, O, n, 1)
    out["outlier_reactors_vs_reference"][f"n{n}"] = {
        "reactors": int(S), "steps": steps, "compared_every": every,
        "gpu_max_rel_err": float(e_gpu.max()), "gpu_reactors_beyond_1e-6": int((e_gpu > 1e-6).sum()),
        "gpu_checkpoints_with_reference_counters": same, "checkpoints": int(S * (steps // every)),
        "oracle_dense_lu_max_rel_err": float(e_lu.max()), "oracle_dense_lu_reactors_beyond_1e-6": int((e_lu > 1e-6).sum()),
        "oracle_tridiagonal_max_rel_err": float(e_tri.max()), "oracle_tridiagonal_reactors_beyond_1e-6": int((e_tri > 1e-6).sum())}
print(json.dumps(out["outlier_reactors_vs_reference"], indent=1))

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out["bench_ensemble_vs_oracle"].items()}, indent=1))
worst_g = max(max(v["max_rel_pH"], v["max_rel_Cl"], v["max_rel_T"]) for v in out["golden_trajectories"].values())
print("golden trajectories: worst rel err", worst_g, "counter mismatches",
      sum(len(v["steps_with_different_solver_counters"]) for v in out["golden_trajectories"].values()))
