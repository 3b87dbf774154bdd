"""In-library multi-stream schedule probe (developer tool)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
STEPS = 500
cols, bc = wt.make_ensemble(N)
for S, chunk, timing in ((1, 50, False), (4, 10, False), (4, 10, True), (2, 10, False), (3, 10, False), (4, 5, False), (8, 10, False), (4, 1, False), (4, 25, False)):
    ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
    ens.set_schedule(S, chunk)
    ens.step(1.0, n_steps=100, download=False); ens.synchronize()
    ens.launch_timing(timing)
    t0 = time.perf_counter()
    ens.step(1.0, n_steps=STEPS, download=False); ens.synchronize()
    dt = time.perf_counter() - t0
    extra = ""
    if timing:
        nl, sm, mx = ens.launch_stats(); extra = f" launches {nl} avg {sm/nl*1e3:.0f} us in-flight {sm/(dt*1e3):.2f}"
    print(f"streams {S} chunk {chunk:3d} timing {timing}: {dt/STEPS*1e6:7.1f} us/step {N*n*STEPS/dt:.3e} zs/s{extra}", flush=True)
    ens.close()
