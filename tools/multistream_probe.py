"""Does splitting the ensemble over several handles (= HIP streams) with short
launches let the hardware queues pack wavefronts better? (developer probe)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
n, N, STEPS = 8, 10000, 500
for S, chunk in ((1, 50), (1, 500), (2, 5), (4, 5), (4, 10), (4, 1), (8, 5), (8, 1), (16, 5), (5, 5), (5, 1)):
    enss = []
    for s in range(S):
        lo, hi = wt.shard_bounds(N, S, s)
        cols, bc = wt.make_ensemble(hi - lo, start=lo)
        e = wt.ReactorEnsemble(cols, n_zones=n); e.set_boundary(bc)
        e.step(1.0, n_steps=100, fused=True, download=False)
        enss.append(e)
    for e in enss: e.synchronize()
    t0 = time.perf_counter()
    for c in range(STEPS // chunk):
        for e in enss:
            e.step(1.0, n_steps=chunk, fused=True, download=False)
    for e in enss: e.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams {S:2d} chunk {chunk:3d}: {dt/STEPS*1e6:7.1f} us/step  {N*n*STEPS/dt:.3e} zone-steps/s", flush=True)
    for e in enss: e.close()
