"""Quick GPU-vs-oracle comparison on the synthetic ensemble (developer tool)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
import wt_oracle as O

def run(n, N, steps, fused):
    cols, bc = wt.make_ensemble(N)
    ens = wt.ReactorEnsemble(cols, n_zones=n)
    ens.set_boundary(bc)
    st0 = ens.state
    # RHS parity
    dp, dc, dT, fl = ens.derivatives(st0.pH, st0.chlorine, st0.temperature)
    par = ens.constants
    worst = 0
    for r in range(min(N, 64)):
        y = np.concatenate([st0.pH[r], st0.chlorine[r], st0.temperature[r]])
        fo, _ = O.rhs(n, par[:, r], bc[:, r], y)
        fg = np.concatenate([dp[r], dc[r], dT[r]])
        worst = max(worst, np.max(np.abs(fo - fg) / np.maximum(np.abs(fo), 1e-12)))
    print(f"n={n} rhs worst rel (vs |f|, floor 1e-12): {worst:.3e}")
    pH, Cl, T, t = st0.pH.copy(), st0.chlorine.copy(), st0.temperature.copy(), st0.time.copy()
    t0 = time.time()
    if fused:
        es = ens.step(1.0, n_steps=steps, fused=True)
        pH, Cl, T, t, ost = O.ensemble_step(n, par, bc, 1.0, steps, pH, Cl, T, t, nthreads=8)
        chk(n, es, pH, Cl, T, ost, ens)
    else:
        for k in range(steps):
            es = ens.step(1.0, n_steps=1)
            pH, Cl, T, t, ost = O.ensemble_step(n, par, bc, 1.0, 1, pH, Cl, T, t, nthreads=8)
            if k in (0, 1, steps - 1):
                print(f" step {k}:", end=""); chk(n, es, pH, Cl, T, ost, ens)
    print(f"   wall {time.time()-t0:.2f}s")

def chk(n, es, pH, Cl, T, ost, ens):
    rel = lambda a, b: np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))
    st = ens.solver_stats()
    print(f" n={n} maxrel pH {rel(es.pH,pH):.2e} Cl {rel(es.chlorine,Cl):.2e} T {rel(es.temperature,T):.2e} "
          f"status gpu {np.bincount(es.status.astype(int)).tolist()} oracle {np.bincount(ost).tolist()} "
          f"mean nfev {st[:,0].mean():.2f} max {st[:,0].max()} njev {st[:,1].mean():.2f} nlu {st[:,2].mean():.2f} steps {st[:,3].mean():.2f} rej {st[:,4].mean():.3f}")

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    for n in (4, 8, 20, 5):
        run(n, N, steps, fused=False)
        run(n, N, steps, fused=True)
