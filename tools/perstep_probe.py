"""Per-step launches (one PLC scan per outer step): throughput vs number of reactor ranges, without per-launch events."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
n, N = 8, 10000
cols, bc = wt.make_ensemble(N)
for io in (False, True):
    for S in (1, 2, 3, 4, 6, 8):
        ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
        if io:
            ens.enable_sensors(seed=1); ens.enable_plant_io(); ens.write_commands(bc[4], bc[6], bc[0])
        ens.set_schedule(S, 1)
        ens.step(1.0, n_steps=100, download=False); ens.synchronize()
        t0 = time.perf_counter(); ens.step(1.0, n_steps=300, download=False); ens.synchronize(); dt = time.perf_counter() - t0
        print(f"plant_io={io} ranges={S}: {N*n*300/dt:.3e} zone-steps/s, {dt/300*1e6:.1f} us/step")
        ens.close()
